cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6p; mkdir -p $O
M=$PWD/fair_marl_amd/csrc/variants/libfmarl_measure.so
FMARL_LIB=$M timeout -k 10 200 python tools/phase_ticks.py fnav 0.5 600 2>&1 | grep -v libdrm > $O/ticks_fnav_steady.txt; cat $O/ticks_fnav_steady.txt
FMARL_TICKS_SPAN=24 FMARL_LIB=$M timeout -k 10 200 python tools/phase_ticks.py fnav 0.5 600 2>&1 | grep -v libdrm > $O/ticks_fnav_steady_span.txt; cat $O/ticks_fnav_steady_span.txt
