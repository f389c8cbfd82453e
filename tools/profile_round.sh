#!/bin/bash
# Collects the evidence bench.py's numbers rest on, on the GPU box:  tools/profile_round.sh <tag> [config]   (e.g. r02a C3x80)
#   gpurun_out/<tag>/bench_<config>_mixed.json                 the bench line (config C3: the default command AND the driver's --steps 20 --warmup 5)
#   gpurun_out/<tag>/kernel_stats_<config>.csv                 rocprofv3 --kernel-trace --stats of the same command (shorter run, no secondary blocks)
#   gpurun_out/<tag>/pmc_<config>.json                         HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (separate passes, guide's corrections)
# Copy what should be judged into profiles/ afterwards (profiles/pmc_latest[_<config>].json is what bench.py reads for roofline.traffic).
set -u
TAG=${1:-r02x}
CFG=${2:-C3}
EXTRA=${EXTRA:-}          # extra bench.py flags, e.g. EXTRA=--hbonds (then SUF=_hbonds names the files)
SUF=${SUF:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ -n "${ONLY_PMC:-}" ]; then          # counters again for an existing set (the bench lines and kernel stats of $OUT stay)
    if [[ "$CFG" != C3x* ]]; then PSTEPS="--steps 400 --warmup 100"; BS="--steps 2000 --warmup 200"; else PSTEPS="--steps 40 --warmup 10"; BS="--steps 100 --warmup 20"; fi
    # (a short bench line of its own: pmc_summary.py takes the algorithmic bytes from it)
    python3 "$ROOT/bench.py" --config "$CFG" $EXTRA --large-n none $BS --no-cpu-baseline --no-rocprof 2>"$OUT/bench_$CFG.stderr" | tail -1 > "$OUT/bench_${CFG}${SUF}_mixed.json"
elif [ "$CFG" = "C3" ]; then
    python3 "$ROOT/bench.py" $EXTRA 2>"$OUT/bench_$CFG.stderr" | tail -1 > "$OUT/bench_${CFG}${SUF}_mixed.json"
    python3 "$ROOT/bench.py" $EXTRA --gpus 1 --steps 20 --warmup 5 2>>"$OUT/bench_$CFG.stderr" | tail -1 > "$OUT/bench_${CFG}${SUF}_mixed_driver_flags.json"
    STEPS="--steps 4000 --warmup 400"; PSTEPS="--steps 400 --warmup 100"
elif [[ "$CFG" != C3x* ]]; then          # the other BASELINE configurations (C1, C2, C4, C5): long run + the driver's flags
    python3 "$ROOT/bench.py" --config "$CFG" $EXTRA --large-n none --no-cpu-baseline 2>"$OUT/bench_$CFG.stderr" | tail -1 > "$OUT/bench_${CFG}${SUF}_mixed.json"
    python3 "$ROOT/bench.py" --config "$CFG" $EXTRA --large-n none --no-cpu-baseline --gpus 1 --steps 20 --warmup 5 2>>"$OUT/bench_$CFG.stderr" | tail -1 > "$OUT/bench_${CFG}${SUF}_mixed_driver_flags.json"
    STEPS="--steps 4000 --warmup 400"; PSTEPS="--steps 400 --warmup 100"
else
    python3 "$ROOT/bench.py" --config "$CFG" $EXTRA --large-n none --steps 200 --warmup 40 --no-cpu-baseline 2>"$OUT/bench_$CFG.stderr" | tail -1 > "$OUT/bench_${CFG}${SUF}_mixed.json"
    STEPS="--steps 100 --warmup 20"; PSTEPS="--steps 40 --warmup 10"
fi
[ -z "${ONLY_PMC:-}" ] && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$CFG" -o run -- python3 "$ROOT/bench.py" --config "$CFG" $EXTRA --large-n none $STEPS --no-cpu-baseline --no-rocprof --no-other-configs > "$OUT/stats_$CFG.log" 2>&1
[ -z "${ONLY_PMC:-}" ] && cp "$OUT/stats_$CFG"/run_kernel_stats.csv "$OUT/kernel_stats_$CFG$SUF.csv" 2>/dev/null || find "$OUT/stats_$CFG" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_$CFG$SUF.csv" \;
for C in FETCH_SIZE WRITE_SIZE; do
    # (--headline-only: the secondary blocks launch OTHER variants of the kernels -- the constrained box's -- more often than the headline's)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o run -- python3 "$ROOT/bench.py" --config "$CFG" $EXTRA --large-n none $PSTEPS --no-cpu-baseline --no-rocprof --headline-only > "$OUT/pmc_${CFG}_$C.log" 2>&1
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" "$TAG" "$CFG" > "$OUT/pmc_$CFG$SUF.json"
rm -rf "$OUT/stats_$CFG" "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE
cat "$OUT/bench_${CFG}${SUF}_mixed.json"; head -6 "$OUT/kernel_stats_$CFG$SUF.csv"; cat "$OUT/pmc_$CFG$SUF.json"
