#!/bin/bash
# Two (or more) builds of the library on ONE box, in rotation, one-launch step of each configuration:  tools/probes/ab_lib_rates.sh "C3 C5 C2 C1" lib1.so lib2.so
CFGS=$1; shift
for cfg in $CFGS; do
  for i in $(seq 1 ${ROUNDS:-3}); do
    for l in "$@"; do
      echo -n "$(basename $l) "; VVHIP_LIB=$PWD/$l python tools/probes/fused_one.py $cfg 1 20000 2>/dev/null | tail -1
    done
  done
done
