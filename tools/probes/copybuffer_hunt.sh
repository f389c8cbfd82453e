cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/cb; rocprofv3 --kernel-trace --output-format csv -d /tmp/cb -o run -- python3 $R/bench.py --config C3x8 --steps 100 --warmup 20 --no-cpu-baseline --no-rocprof --large-n none > /tmp/cb.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/cb/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][-40:] for r in rows]
print(collections.Counter(names).most_common(8))
idx = [i for i, n in enumerate(names) if "copyBuffer" in n]
print("copyBuffer at", idx[:40], "...", len(idx))
for i in idx[10:14]:
    print([names[j][-22:] for j in range(max(0, i - 3), min(len(names), i + 3))])
PY
