cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cfg in C3 C2 C5; do for f in 1 0; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${cfg}_$f -o x -- python3 $R/tools/probes/fused_one.py $cfg $f 4000 2>&1 | grep "steps/s"
python3 -c "
import csv,glob
for f in glob.glob('/tmp/prof_${cfg}_$f/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print('   ', r['Name'][:64], r['Calls'], r['AverageNs'])
"
done; done
