#!/bin/bash
# A variant of the library for A/B runs on one box (tools/probes/ab_lib_rates.sh):  tools/probes/build_variant.sh <name> [-DFLAG=VALUE ...]
# Only the one-launch step's object (vv_kernels.hip, VV_KERNELS_PART=2) is recompiled with the flags; the other objects are the product build's
# (make -C openmm-velocityverlet_amd/csrc first).  PART=1 recompiles the two-launch kernels' object instead, PART="1 2" both.  Result: tools/probes/libs/libvvhip_<name>.so
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/openmm-velocityverlet_amd/csrc; OBJ=$ROOT/openmm-velocityverlet_amd/lib/obj; OUT=$ROOT/tools/probes/libs
mkdir -p "$OUT/obj_$NAME"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I/opt/rocm/include -mllvm -amdgpu-kernarg-preload-count=16"
OBJS="$OBJ/vv_host.o $OBJ/vv_api.o $OBJ/vv_rtc.o"
for part in 1 2; do
  if [[ " ${PART:-2} " == *" $part "* ]]; then
    /opt/rocm/bin/hipcc $FLAGS "$@" -DVV_KERNELS_PART=$part -c -o "$OUT/obj_$NAME/vv_kernels_$part.o" "$SRC/vv_kernels.hip" &
    OBJS="$OBJS $OUT/obj_$NAME/vv_kernels_$part.o"
  else
    OBJS="$OBJS $OBJ/vv_kernels_$part.o"
  fi
done
wait
/opt/rocm/bin/hipcc $FLAGS -shared -o "$OUT/libvvhip_$NAME.so" $OBJS -ldl
echo "$OUT/libvvhip_$NAME.so"
