#!/bin/bash
# Where do the kernel arguments live?  HIP_FORCE_DEV_KERNARG=1 puts the kernarg segment into device memory instead of host memory: every kernel's
# scalar loads of its arguments (KArgs is 1.4 KB; only the first 16 dwords are preloaded into SGPRs) then hit HBM / L2 instead of crossing PCIe.
# steps/s of the one-launch step per configuration, unset / 0 / 1 in rotation on one box:   tools/probes/kernarg_env.sh "C3 C2 C5"
for cfg in ${1:-C3 C2}; do
  for i in $(seq 1 ${ROUNDS:-3}); do
    echo -n "unset "; env -u HIP_FORCE_DEV_KERNARG python tools/probes/fused_one.py $cfg 1 20000 2>/dev/null | tail -1
    echo -n "HIP_FORCE_DEV_KERNARG=0 "; HIP_FORCE_DEV_KERNARG=0 python tools/probes/fused_one.py $cfg 1 20000 2>/dev/null | tail -1
    echo -n "HIP_FORCE_DEV_KERNARG=1 "; HIP_FORCE_DEV_KERNARG=1 python tools/probes/fused_one.py $cfg 1 20000 2>/dev/null | tail -1
  done
done
