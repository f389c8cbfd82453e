#!/bin/bash
# (every run under its own timeout: ROC_SYSTEM_SCOPE_SIGNAL=0 never came back, round 6)
# The HIP runtime's own switches around a graph replay of the one-launch step (nothing the library can set: a host program's environment), in rotation on ONE box:
#   bash tools/probes/runtime_envs.sh "C3 C2" "-" "AMD_OPT_FLUSH=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" ...        ("-" = the defaults)
CFGS=$1; shift
for cfg in $CFGS; do
  for i in $(seq 1 ${ROUNDS:-2}); do
    for e in "$@"; do
      if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
      echo -n "[$e] "; env $ee timeout ${PER_RUN_TIMEOUT:-90} python tools/probes/fused_one.py $cfg 1 20000 2>/dev/null | tail -1; echo
    done
  done
done
