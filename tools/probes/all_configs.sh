#!/bin/bash
# One line per BASELINE configuration (mixed precision, same box): tools/probes/all_configs.sh > gpurun_out/<tag>/all_configs.txt
for cfg in C1 C2 C3 C4 C5 C3x8 C3x80; do
  case $cfg in C3x80) ST="--steps 200 --warmup 40";; C3x8) ST="--steps 2000 --warmup 200";; *) ST="--steps 20000 --warmup 2000";; esac
  python bench.py --config $cfg --large-n none $ST --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; w=c.get('with_constraints') or {}
print('$cfg', c['workload'].split(';')[0], '| steps/s', d['value'], '| integrator only', c.get('integrator_only_steps_per_s'), '| constrained', w.get('steps_per_s'), '(driver flags', w.get('steps_per_s_driver_flags'), ') | A/B us', r['avg_launch_us'], '| dominant', r['kernel'], r['frac'])"
done
for cfg in C3 C4 C5; do      # the driver's flags: one replay of a 20-step graph per timed region
  python bench.py --config $cfg --large-n none --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rocprof 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg under the driver flags (--steps 20 --warmup 5): steps/s', d['value'], d['config']['timed_region_ms'])"
done
for prec in single double; do
  python bench.py --precision $prec --large-n none --steps 4000 --warmup 400 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('C3 $prec', d['value'], c.get('integrator_only_steps_per_s'), d['roofline']['avg_launch_us'])"
done
python bench.py --synthetic --large-n none --steps 4000 --warmup 400 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('C3 synthetic (round-1 look-alike)', d['value'], c.get('integrator_only_steps_per_s'), (c.get('with_constraints') or {}).get('steps_per_s'), d['roofline']['avg_launch_us'])"
