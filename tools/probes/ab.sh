#!/bin/bash
# A/B comparison of two builds of libvvhip on ONE GPU box (boxes differ by up to ~12 %): tools/probes/ab.sh <variant.so> [bench args]
# alternates baseline (the in-tree lib) and variant three times each and prints value / A / B / large-N figures per run.
V=$1; shift
L=openmm-velocityverlet_amd/lib/libvvhip.so
cp $L /tmp/base.so
for i in 1 2 3; do
  for which in base variant; do
    if [ $which = base ]; then cp /tmp/base.so $L; else cp $V $L; fi
    python bench.py --steps 4000 --warmup 400 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); ln=d['config'].get('large_n'); print('$which', d['value'], d['roofline']['avg_launch_us'], {k:v['avg_launch_us'] for k,v in ln['roofline'].items()} if ln else '')"
  done
done
cp /tmp/base.so $L
