"""HBM traffic per launch of the fused kernels from rocprofv3 counter CSVs (tools/profile_round.sh).
Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes for gfx950: FETCH_SIZE and WRITE_SIZE are
in KB; FETCH_SIZE reads exactly half of a wide coalesced streaming read -> doubled; WRITE_SIZE as is."""
import csv, glob, json, os, sys

out_dir, tag = sys.argv[1], sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else "C3"
natoms = {"C1": 1992, "C2": 9999, "C3": 111000, "C4": 111000, "C5": 40310, "C3x8": 888000, "C3x80": 8880000}.get(cfg, 111000)
names = {"vv_kernel_a": "A", "vv_kernel_b": "B", "vv_kernel_tether": "tether"}
raw, variants, chosen = {}, {}, {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out_dir, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            for key, short in names.items():
                if key in row["Kernel_Name"]:
                    # one bench run launches several compiled variants of a kernel (headline stage set, the constrained box of the
                    # secondary figure, ...): keep them apart by their full template name, report the most-launched one below
                    rec = variants.setdefault(short, {}).setdefault(row["Kernel_Name"], {}).setdefault(counter, [0.0, 0])
                    rec[0] += float(row["Counter_Value"]); rec[1] += 1
for short, byname in variants.items():
    name = max(byname, key=lambda k: max(v[1] for v in byname[k].values()))
    raw[short], chosen[short] = byname[name], name
res = {"config": cfg, "precision": "mixed", "round": tag,
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --config %s --large-n none --no-cpu-baseline (two separate passes; tools/profile_round.sh)" % cfg,
       "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE (KB) reads exactly half of a wide coalesced streaming read on gfx950 -> doubled; WRITE_SIZE (KB) as is; x1024 for bytes",
       "kernel_variant": chosen, "raw_kb": {}}
for short, d in raw.items():
    res["raw_kb"][short] = {c: round(v[0] / max(v[1], 1), 1) for c, v in d.items()}
    res["raw_kb"][short]["dispatches"] = max(v[1] for v in d.values())
    f = d.get("FETCH_SIZE", [0, 1]); w = d.get("WRITE_SIZE", [0, 1])
    res[f"hbm_bytes_per_launch_{short}"] = int(round((2 * f[0] / max(f[1], 1) + w[0] / max(w[1], 1)) * 1024))
# algorithmic bytes per particle: what the plan itself reports (vvhip_algorithmic_bytes, printed by bench.py as
# roofline.algorithmic_bytes_per_particle) for the same configuration -- taken from the bench line of this profiling round
suf = os.environ.get("SUF", "")
per, src = None, None
for name in (f"bench_{cfg}{suf}_mixed.json", f"bench_{cfg}_mixed.json"):
    try:
        b = json.load(open(os.path.join(out_dir, name)))
        blk = b["roofline"] if b["config"]["workload"].startswith(cfg + ":") or not cfg.startswith("C3x") else b["roofline"]
        per, src = blk["algorithmic_bytes_per_particle"], name
        break
    except Exception:
        continue
if per is None:      # no bench line at hand: the headline path's figures (mixed precision, register-kick path / store path)
    per, src = ({"A": 62, "B": 158} if cfg.startswith("C3") or cfg in ("C1", "C2") else {"A": 94, "B": 134}), "built-in default"
res["algorithmic_bytes_per_particle"] = per
res["algorithmic_bytes_source"] = src
res["algorithmic_bytes_per_launch"] = {k: v * natoms for k, v in per.items()}
for k in ("A", "B"):
    if f"hbm_bytes_per_launch_{k}" in res:
        res[f"traffic_over_algorithmic_{k}"] = round(res[f"hbm_bytes_per_launch_{k}"] / res["algorithmic_bytes_per_launch"][k], 3)
print(json.dumps(res, indent=1))
