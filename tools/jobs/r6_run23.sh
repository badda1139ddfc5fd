cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6x; mkdir -p $O
M=$PWD/fair_marl_amd/csrc/variants/libfmarl_measure.so
FMARL_TICKS_SPAN=24 FMARL_LIB=$M timeout -k 10 200 python tools/phase_ticks.py cfg4 2>&1 | grep -v libdrm > $O/ticks_cfg4_span.txt; cat $O/ticks_cfg4_span.txt
FMARL_TICKS_SPAN=24 FMARL_LIB=$M timeout -k 10 200 python tools/phase_ticks.py n10 2>&1 | grep -v libdrm > $O/ticks_n10_span.txt; cat $O/ticks_n10_span.txt
