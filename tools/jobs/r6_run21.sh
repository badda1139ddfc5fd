cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6v; mkdir -p $O
M=$PWD/fair_marl_amd/csrc/variants/libfmarl_measure.so
FMARL_LIB=$M timeout -k 10 200 python tools/phase_ticks.py fnav10 2>&1 | grep -v libdrm > $O/ticks_fnav10_step.txt; cat $O/ticks_fnav10_step.txt
