cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s; mkdir -p $O
M=$PWD/fair_marl_amd/csrc/variants/libfmarl_measure.so
FMARL_LIB=$M timeout -k 10 300 python tools/phase_ticks.py cfg3 2>&1 | grep -v libdrm > $O/ticks_cfg3_step.txt; cat $O/ticks_cfg3_step.txt
