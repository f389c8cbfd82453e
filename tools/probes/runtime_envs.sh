#!/bin/bash
# The HIP runtime's own switches around a graph replay of the one-launch step (nothing the library can set: a host program's environment), in rotation on ONE box:
#   bash tools/probes/runtime_envs.sh "C3 C2" "-" "AMD_OPT_FLUSH=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" ...        ("-" = the defaults)
CFGS=$1; shift
for cfg in $CFGS; do
  for i in $(seq 1 ${ROUNDS:-2}); do
    for e in "$@"; do
      if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
      echo -n "[$e] "; env $ee python tools/probes/fused_one.py $cfg 1 20000 2>/dev/null | tail -1
    done
  done
done
