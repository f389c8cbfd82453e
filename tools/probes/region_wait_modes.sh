#!/bin/bash
# The fixed cost of the driver's 20-step timed region under the HIP runtime's host-wait knobs (tools/probes/region_overhead.py each time)
for v in "" "ROC_ACTIVE_WAIT_TIMEOUT=0" "ROC_ACTIVE_WAIT_TIMEOUT=100" "ROC_ACTIVE_WAIT_TIMEOUT=100000" "ROC_CPU_WAIT_FOR_SIGNAL=0" "ROC_CPU_WAIT_FOR_SIGNAL=1" "ROC_CPU_WAIT_FOR_SIGNAL=1 ROC_ACTIVE_WAIT_TIMEOUT=100000"; do
  echo "== ${v:-default}"
  env $v python tools/probes/region_overhead.py 2>/dev/null | grep -v amdgpu.ids
done
