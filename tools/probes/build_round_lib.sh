#!/bin/bash
# The library of an EARLIER commit, for same-box A/B runs against the present one (tools/probes/ab_lib_rates.sh; boxes differ by up to 12 %, so a
# round's gain is only ever quoted from two builds in rotation on one box):   tools/probes/build_round_lib.sh <commit> <name>
# builds tools/probes/libs/libvvhip_<name>.so from that commit's openmm-velocityverlet_amd/csrc + include (build container only: needs .git).
# The Python side tolerates entry points the old library lacks when it is named through VVHIP_LIB (vvhip.py: _load).
set -eu
C=$1; N=$2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d /tmp/vvround.XXXX)
git -C "$ROOT" archive "$C" openmm-velocityverlet_amd/csrc include | tar -x -C "$T"
make -s -C "$T/openmm-velocityverlet_amd/csrc" > "$T/make.log" 2>&1 || { tail -5 "$T/make.log"; exit 1; }
mkdir -p "$ROOT/tools/probes/libs"
cp "$T/openmm-velocityverlet_amd/lib/libvvhip.so" "$ROOT/tools/probes/libs/libvvhip_$N.so"
rm -rf "$T"
echo "$ROOT/tools/probes/libs/libvvhip_$N.so"
