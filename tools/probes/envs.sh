#!/bin/bash
# The same build under several environment settings on ONE GPU box:  tools/probes/envs.sh "<bench args>" "VAR=1" "VAR=0 OTHER=2" ...
# ("-" = no extra variables); every setting ROUNDS times in rotation.
ARGS=$1; shift
for i in $(seq 1 ${ROUNDS:-2}); do
  for e in "$@"; do
    if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
    env $ee python bench.py $ARGS --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$e]', d['value'], d['config'].get('integrator_only_steps_per_s'), d['roofline']['avg_launch_us'], d['roofline']['frac'])"
  done
done
