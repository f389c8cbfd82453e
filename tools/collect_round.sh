set -u
R=$GRAFT_REPO_ROOT; T=r05h; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
VVHIP_LIB=$R/tools/probes/libs/libvvhip_ts.so python tools/probes/fused_timeline.py C3 C4 C5 C2 > $O/fused_timeline.txt 2>&1
for f in 1 0; do FUSED=$f VVHIP_LIB=$R/tools/probes/libs/libvvhip_ts.so python tools/probes/step_anatomy.py C3 C4 C5 C2 > $O/step_anatomy_fused$f.txt 2>&1; done
timeout 900 python tools/probes/fused_ab.py C3,C4,C5,C2,C1,C3hb,C5hb,C2hb 3 > $O/fused_ab_all_configs.txt 2>&1
bash tools/profile_round.sh $T C3 > /dev/null 2>&1
bash tools/profile_round.sh $T C4 > /dev/null 2>&1
EXTRA=--hbonds SUF=_hbonds bash tools/profile_round.sh $T C3 > /dev/null 2>&1
bash tools/profile_round.sh $T C5 > /dev/null 2>&1
bash tools/profile_round.sh $T C3x80 > /dev/null 2>&1
bash tools/pmc_sq.sh $T C3 > /dev/null 2>&1
bash tools/pmc_sq.sh $T C4 > /dev/null 2>&1
bash tools/pmc_sq.sh $T C3x80 > /dev/null 2>&1
ls $O
