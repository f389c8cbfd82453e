#!/bin/bash
# Begin / end timestamps of consecutive dispatches from rocprofv3's kernel trace, graph replay vs eager launches (bench.py [--eager]):
# how long each kernel runs and how large the gaps between them are -- what "duration in sequence" means for a 5 us kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
EXTRA=${EXTRA:-}
for mode in graph eager; do
  EX=""; [ $mode = eager ] && EX="--eager"
  rm -rf /tmp/tl_$mode
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$mode -o run -- python3 $R/bench.py $EXTRA --large-n none --steps 2000 --warmup 200 --no-cpu-baseline --no-rocprof $EX > /tmp/tl_$mode.log 2>&1
  echo "== $mode: $(tail -1 /tmp/tl_$mode.log | cut -c1-120)"
  python3 - /tmp/tl_$mode <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"vv_kernel_(\w+)<[^,]*,[^,]*(?:, (\d+)u)?", n)
    return (m.group(1) + ("/" + m.group(2) if m.group(2) else "")) if m else n[:20]
# a window in the middle of the timed run
mid = len(rows) // 2
import statistics
dur, gap = {}, {}
for i in range(len(rows) // 4, 3 * len(rows) // 4):
    k = short(rows[i]["Kernel_Name"])
    dur.setdefault(k, []).append(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]))
    gap.setdefault(k, []).append(int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]))
for k in dur:
    if len(dur[k]) > 50:
        print("  %-22s n=%5d  duration median %6.0f ns  mean %6.0f   gap to predecessor's end: median %6.0f ns  mean %6.0f" % (k, len(dur[k]), statistics.median(dur[k]), statistics.mean(dur[k]), statistics.median(gap[k]), statistics.mean(gap[k])))
t0 = int(rows[mid]["Start_Timestamp"])
for r in rows[mid:mid + 7]:
    print("    %-22s start %7d end %7d" % (short(r["Kernel_Name"]), int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0))
PY
done
