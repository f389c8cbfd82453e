#!/bin/bash
# This tree against another checkout of the repository (e.g. last round's, copied to tools/probes/r02tree with its built library) on ONE box:
#   tools/probes/ab_trees.sh <other tree> "<bench args>" [rounds]
O=$1; ARGS=$2
for i in $(seq 1 ${3:-2}); do
  for which in other this; do
    if [ $which = other ]; then B=$O/bench.py; X=""; else B=bench.py; X="--no-rocprof"; fi
    python $B $ARGS $X --no-cpu-baseline --large-n none 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$which', d['value'], c.get('integrator_only_steps_per_s'), (c.get('with_constraints') or {}).get('steps_per_s'), d['roofline']['avg_launch_us'])"
  done
done
