#!/bin/bash
# Several builds of libvvhip on ONE GPU box (boxes differ by up to ~12 %):  tools/probes/variants.sh "<bench args>" base var1.so var2.so ...
# runs every build ROUNDS times in rotation and prints value / kernel A / kernel B average launch times per run.
ARGS=$1; shift
L=openmm-velocityverlet_amd/lib/libvvhip.so
cp $L /tmp/base.so
for i in $(seq 1 ${ROUNDS:-2}); do
  for v in "$@"; do
    if [ "$v" = base ]; then cp /tmp/base.so $L; else cp "$v" $L; fi
    python bench.py $ARGS --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $v)', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
  done
done
cp /tmp/base.so $L
