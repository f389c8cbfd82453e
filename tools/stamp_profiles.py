"""Copies the evidence of a profiling round from gpurun_out/<tag>/ into profiles/ (tracked) and refreshes what bench.py reads:
   profiles/pmc_latest[_<config>][_hbonds].json   HBM bytes per launch (PMC), stamped with the commit they were measured at
   profiles/kernel_stats_latest.json              rocprofv3 --kernel-trace --stats averages per kernel variant, per workload
Usage: python tools/stamp_profiles.py <tag> [<tag> ...]      (run in the build container, where .git exists)"""
import csv, glob, json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
if os.environ.get("STAMP_COMMIT"):      # the set was measured at an earlier commit than HEAD (say which)
    commit = os.environ["STAMP_COMMIT"]
dirty = not os.environ.get("STAMP_COMMIT") and bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "openmm-velocityverlet_amd", "bench.py"], capture_output=True, text=True).stdout.strip())
stats_path = os.path.join(ROOT, "profiles", "kernel_stats_latest.json")
stats = json.load(open(stats_path)) if os.path.exists(stats_path) else {}
for tag in sys.argv[1:]:
    src = os.path.join(ROOT, "gpurun_out", tag)
    for f in sorted(glob.glob(os.path.join(src, "*"))):
        base = os.path.basename(f)
        if os.path.isdir(f) or base.endswith(".log") or base.endswith(".stderr"):
            continue
        m = re.match(r"(bench|kernel_stats|pmc_sq|pmc)_(C\d+(?:x\d+)?)(_hbonds)?(.*)\.(json|csv)$", base)
        if not m:
            shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_{base}"))
            continue
        kind, cfg, hb, rest, ext = m.groups()
        hb = hb or ""
        names = {"bench": f"{tag}_bench_{cfg}{hb}_mixed{rest.replace('_mixed', '')}.json", "kernel_stats": f"{tag}_rocprofv3_kernel_stats_bench_{cfg}{hb}_mixed.csv",
                 "pmc": f"{tag}_pmc_{cfg}{hb}_mixed.json", "pmc_sq": f"{tag}_pmc_sq_{cfg}{hb}_mixed.json"}
        dst = os.path.join(ROOT, "profiles", names[kind])
        if kind in ("pmc", "pmc_sq") and os.path.getsize(f) > 2:
            rec = json.load(open(f))
            rec["commit"] = commit + ("+uncommitted" if dirty else "") if commit else None
            rec["round"] = tag
            json.dump(rec, open(dst, "w"), indent=1)
            if kind == "pmc":
                latest = "pmc_latest" + ("" if cfg == "C3" else "_" + cfg) + hb + ".json"
                json.dump(rec, open(os.path.join(ROOT, "profiles", latest), "w"), indent=1)
        else:
            shutil.copy(f, dst)
        if kind == "kernel_stats":
            rows = {}
            for r in csv.DictReader(open(f)):
                mm = re.match(r"void vv::(vv_kernel_\w+)<([^>]*)>", r["Name"])
                if mm:
                    rows[f"{mm.group(1)}<{mm.group(2)}>"] = {"calls": int(r["Calls"]), "avg_ns": round(float(r["AverageNs"]), 1)}
            stats[cfg + hb] = {"file": "profiles/" + names[kind], "round": tag, "commit": commit + ("+uncommitted" if dirty else "") if commit else None,
                               "command": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --config {cfg}{' --hbonds' if hb else ''} --large-n none --no-cpu-baseline (graph replay; tools/profile_round.sh)",
                               "kernels": rows}
json.dump(stats, open(stats_path, "w"), indent=1)
print("profiles/ refreshed from", sys.argv[1:], "at commit", commit, "(+uncommitted)" if dirty else "")
