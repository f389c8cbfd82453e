cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in graph eager; do
  EX=""; [ $mode = eager ] && EX="--eager"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$mode -o run -- python3 $R/bench.py --large-n none --steps 4000 --warmup 400 --no-cpu-baseline --no-rocprof $EX > /tmp/st_$mode.log 2>&1
  echo "== $mode"; tail -1 /tmp/st_$mode.log | cut -c1-200; find /tmp/st_$mode -name "*kernel_stats.csv" -exec head -5 {} \; 
done
python3 $R/tools/probes/dispatch_timing.py | head -3
