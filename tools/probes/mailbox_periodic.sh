#!/bin/bash
# Round-4 experiment: the mailbox exchange next to kernel B's arithmetic layout, two processes on ONE GPU.
# Variants: layout off (the conservative default when ranks share a device) / forced on with both processes' grids capped so that
# their kernels are resident together / forced on with the default (device-filling) grids.  Prints one line per variant.
R=$(cd "$(dirname "$0")/../.." && pwd)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
port=29610
run() {  # label, env...
    label=$1; shift
    port=$((port+1))
    out=$(env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port $R/tests/dist_worker.py mailbox_periodic 2>&1)
    echo "== $label: $(echo "$out" | grep -E 'MAILBOX PERIODIC|Error|error|assert' | head -3 | tr '\n' ' ')"
}
for cfg in ${CONFIGS:-4 8}; do
  run "C3x$cfg layout off (default on a shared device)"            VV_MBP_CONFIG=$cfg
  run "C3x$cfg layout ON, grids capped 96/96 (co-resident)"         VV_MBP_CONFIG=$cfg VVHIP_PERIODIC_MB=1 VVHIP_CAP_A=96 VVHIP_CAP_B=96
  run "C3x$cfg layout ON, grids capped 128/128"                     VV_MBP_CONFIG=$cfg VVHIP_PERIODIC_MB=1 VVHIP_CAP_A=128 VVHIP_CAP_B=128
  run "C3x$cfg layout ON, default grids"                            VV_MBP_CONFIG=$cfg VVHIP_PERIODIC_MB=1
  run "C3x$cfg layout off, grids capped 96/96"                      VV_MBP_CONFIG=$cfg VVHIP_CAP_A=96 VVHIP_CAP_B=96
done
