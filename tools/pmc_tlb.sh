#!/bin/bash
# Address-translation and L2 hit counters of the fused kernels at the headline size (own rocprofv3 passes, counters only):  tools/pmc_tlb.sh <tag> [config]
set -u
TAG=${1:-r04x}
CFG=${2:-C3}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
P=0
for SET in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum"; do
    P=$((P+1))
    rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_tlb$P" -o run -- python3 "$ROOT/bench.py" --config "$CFG" --large-n none --steps 400 --warmup 100 --no-cpu-baseline --no-rocprof --headline-only > "$OUT/pmc_tlb_${CFG}_$P.log" 2>&1
done
python3 - "$OUT" "$TAG" "$CFG" <<'PY' > "$OUT/pmc_tlb_$CFG.json"
import csv, glob, json, os, re, sys
out, tag, cfg = sys.argv[1:4]
acc = {}
for f in glob.glob(os.path.join(out, "pmc_tlb*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.match(r"void vv::(vv_kernel_\w+)<([^>]*)>", row["Kernel_Name"])
        if not m: continue
        key = f"{m.group(1)}<{m.group(2)}>"
        r = acc.setdefault(key, {}).setdefault(row["Counter_Name"], [0.0, 0])
        r[0] += float(row["Counter_Value"]); r[1] += 1
res = {"round": tag, "config": cfg, "what": "per-launch averages (summed over the GPU)"}
for k, d in sorted(acc.items()):
    res[k] = {"launches": max(v[1] for v in d.values()), "per_launch": {c: round(v[0] / max(v[1], 1), 1) for c, v in d.items()}}
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT"/pmc_tlb1 "$OUT"/pmc_tlb2 "$OUT"/pmc_tlb3
cat "$OUT/pmc_tlb_$CFG.json"; grep -i -c "utcl\|tlb" "$OUT/counters_available.txt"; tail -3 "$OUT"/pmc_tlb_${CFG}_1.log
