#!/bin/bash
# Collects the evidence bench.py's numbers rest on, on the GPU box:  tools/profile_round.sh <tag>   (e.g. r01d)
#   gpurun_out/<tag>/bench_C3_mixed.json            the default bench line
#   gpurun_out/<tag>/kernel_stats.csv               rocprofv3 --kernel-trace --stats of the same command (shorter run)
#   gpurun_out/<tag>/pmc.json                       HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (separate passes, guide's corrections)
# Copy what should be judged into profiles/ afterwards.
set -u
TAG=${1:-r01x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" 2>"$OUT/bench.stderr" | tail -1 > "$OUT/bench_C3_mixed.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/bench.py" --steps 4000 --warmup 400 --no-cpu-baseline > "$OUT/stats.log" 2>&1
cp "$OUT"/stats/run_kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o run -- python3 "$ROOT/bench.py" --steps 400 --warmup 100 --no-cpu-baseline > "$OUT/pmc_$C.log" 2>&1
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" "$TAG" > "$OUT/pmc.json"
rm -rf "$OUT/stats" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE
cat "$OUT/bench_C3_mixed.json"; head -5 "$OUT/kernel_stats.csv"; cat "$OUT/pmc.json"
