#!/bin/bash
# A round's evidence set in one call on the GPU box:  bash tools/collect_round.sh <tag>     (then, in the build container: python tools/stamp_profiles.py <tag>)
set -u
R=$GRAFT_REPO_ROOT; T=${1:-r06x}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
# instrumented build (in-kernel stamps) for the timeline / anatomy probes: built here, where the sources are the ones under test
( cd openmm-velocityverlet_amd/csrc && mkdir -p ../../tools/probes/libs && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I/opt/rocm/include \
    -mllvm -amdgpu-kernarg-preload-count=16 -DVV_KERNEL_TIMESTAMPS -shared -o ../../tools/probes/libs/libvvhip_ts.so vv_host.cpp vv_api.cpp vv_rtc.cpp vv_kernels.hip -ldl > $O/ts_build.log 2>&1 ) &
TS=$!
timeout 900 python tools/probes/fused_ab.py C3,C4,C5,C2,C1,C3hb,C5hb,C2hb 3 > $O/fused_ab_all_configs.txt 2>&1
# this round's kernels against the previous round's on this very box (tools/probes/build_round_lib.sh e195552 r05, in the build container)
[ -f tools/probes/libs/libvvhip_r05.so ] && ROUNDS=3 timeout 600 bash tools/probes/ab_lib_rates.sh "C3 C4 C5 C2 C1 C3hb C5hb C2hb" tools/probes/libs/libvvhip_r05.so openmm-velocityverlet_amd/lib/libvvhip.so > $O/r05_vs_r06_ab.txt 2>&1
CLASSIC=1 timeout 600 python tools/probes/fused_ab.py C3,C4,C5,C2 2 10000 > $O/classic_scheme_ab.txt 2>&1
timeout 600 python tools/probes/shard_step.py C4 8,4,2 > $O/shard_step.txt 2>&1
( time python bench.py > $O/bench_default_timed.json 2> $O/bench_default_timed.stderr ) 2> $O/bench_default_wallclock.txt
bash tools/profile_round.sh $T C3 > /dev/null 2>&1
bash tools/profile_round.sh $T C4 > /dev/null 2>&1
EXTRA=--hbonds SUF=_hbonds bash tools/profile_round.sh $T C3 > /dev/null 2>&1
bash tools/profile_round.sh $T C5 > /dev/null 2>&1
bash tools/profile_round.sh $T C2 > /dev/null 2>&1        # (C2 / C1: bench.py's config.other_configs take their rocprofv3 clock from these when no live child ran)
bash tools/profile_round.sh $T C1 > /dev/null 2>&1
bash tools/profile_round.sh $T C3x80 > /dev/null 2>&1
bash tools/pmc_sq.sh $T C3 > /dev/null 2>&1
bash tools/pmc_sq.sh $T C4 > /dev/null 2>&1
bash tools/pmc_sq.sh $T C3x80 > /dev/null 2>&1
wait $TS
VVHIP_LIB=$R/tools/probes/libs/libvvhip_ts.so python tools/probes/fused_timeline.py C3 C4 C5 C2 > $O/fused_timeline.txt 2>&1
for f in 1 0; do FUSED=$f VVHIP_LIB=$R/tools/probes/libs/libvvhip_ts.so python tools/probes/step_anatomy.py C3 C4 C5 C2 > $O/step_anatomy_fused$f.txt 2>&1; done
rm -f $O/*.log
ls $O
