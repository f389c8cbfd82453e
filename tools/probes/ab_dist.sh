#!/bin/bash
# as ab_libs.sh, for the N = 1 run of the distributed code path (bench.py --force-dist)
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29577
for i in $(seq 1 ${ROUNDS:-2}); do
  for l in "$@"; do
    VVHIP_LIB=$PWD/$l python bench.py --gpus 1 --steps 4000 --warmup 400 --no-cpu-baseline --no-rocprof --large-n none --force-dist --dist-mode mailbox 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$l', d['value'], c['exchange']['chosen'], d['roofline']['avg_launch_us'])"
  done
done
