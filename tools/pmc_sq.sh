#!/bin/bash
# Instruction-mix and issue counters of the fused kernels (own rocprofv3 passes, counters only):  tools/pmc_sq.sh <tag> [config]
set -u
TAG=${1:-r02x}
CFG=${2:-C3}
EXTRA=${EXTRA:-}          # extra bench.py flags, e.g. EXTRA=--hbonds (then SUF=_hbonds names the files)
SUF=${SUF:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [[ "$CFG" != C3x* ]]; then STEPS="--steps 400 --warmup 100"; else STEPS="--steps 40 --warmup 10"; fi
P=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD"; do
    P=$((P+1))
    rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_sq$P" -o run -- python3 "$ROOT/bench.py" --config "$CFG" $EXTRA --large-n none $STEPS --no-cpu-baseline --no-rocprof --headline-only > "$OUT/pmc_sq_${CFG}_$P.log" 2>&1
done
python3 - "$OUT" "$TAG" "$CFG" <<'PY' > "$OUT/pmc_sq_$CFG$SUF.json"
import csv, glob, json, os, re, sys
out, tag, cfg = sys.argv[1:4]
acc = {}
for f in glob.glob(os.path.join(out, "pmc_sq*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.match(r"void vv::(vv_kernel_\w+)<([^>]*)>", row["Kernel_Name"])
        if not m: continue
        key = f"{m.group(1)}<{m.group(2)}>"
        r = acc.setdefault(key, {}).setdefault(row["Counter_Name"], [0.0, 0])
        r[0] += float(row["Counter_Value"]); r[1] += 1
res = {"round": tag, "config": cfg, "precision": "mixed", "what": "per-launch averages of SQ counters (summed over the GPU); per_wave = per wave launched"}
for k, d in sorted(acc.items()):
    per = {c: v[0] / max(v[1], 1) for c, v in d.items()}
    waves = per.get("SQ_WAVES", 0) or 1
    res[k] = {"launches": max(v[1] for v in d.values()), "per_launch": {c: round(x, 1) for c, x in per.items()}, "per_wave": {c: round(x / waves, 1) for c, x in per.items() if c != "SQ_WAVES"}}
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT"/pmc_sq1 "$OUT"/pmc_sq2 "$OUT"/pmc_sq3
cat "$OUT/pmc_sq_$CFG$SUF.json"
