#!/bin/bash
# Several builds of the library on ONE GPU box under the same bench.py:  tools/probes/ab_libs.sh "<bench args>" lib1.so lib2.so ...  (ROUNDS rotations)
ARGS=$1; shift
for i in $(seq 1 ${ROUNDS:-3}); do
  for l in "$@"; do
    VVHIP_LIB=$PWD/$l python bench.py $ARGS --no-rocprof --no-cpu-baseline --large-n none 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$l', d['value'], c.get('integrator_only_steps_per_s'), (c.get('with_constraints') or {}).get('steps_per_s'), d['roofline']['avg_launch_us'])"
  done
done
