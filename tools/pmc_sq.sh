#!/bin/bash
# Instruction-mix counters of the fused kernels (own rocprofv3 pass, counters only):  tools/pmc_sq.sh <tag>
set -u
TAG=${1:-r01x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o run -- python3 "$ROOT/bench.py" --steps 400 --warmup 100 --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY' > "$OUT/pmc_sq.json"
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
names = {"vv_kernel_a<float, double, 1056": "A", "vv_kernel_b<float, double, 2577": "B", "vv_kernel_tether": "tether"}
acc = {}
for f in glob.glob(os.path.join(out, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        for key, short in names.items():
            if key in row["Kernel_Name"]:
                r = acc.setdefault(short, {}).setdefault(row["Counter_Name"], [0.0, 0])
                r[0] += float(row["Counter_Value"]); r[1] += 1
res = {"round": tag, "config": "C3", "precision": "mixed", "what": "per launch averages of SQ counters (summed over the GPU); per_wave = instructions a wave executes",
       "command": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -- python3 bench.py --steps 400 --warmup 100 --no-cpu-baseline"}
for k, d in acc.items():
    per = {c: v[0] / max(v[1], 1) for c, v in d.items()}
    waves = per.get("SQ_WAVES", 0) or 1
    res[k] = {"per_launch": {c: round(x, 1) for c, x in per.items()}, "per_wave": {c: round(x / waves, 1) for c, x in per.items() if c != "SQ_WAVES"}}
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT/pmc_sq"
cat "$OUT/pmc_sq.json"
