set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6d; mkdir -p $O
M=$PWD/fair_marl_amd/csrc/variants/libfmarl_measure.so
FMARL_LIB=$M timeout -k 10 200 python tools/phase_ticks.py fnav10 2>&1 | grep -v libdrm > $O/ticks_fnav10_step.txt; cat $O/ticks_fnav10_step.txt
FMARL_TICKS_SPAN=24 FMARL_LIB=$M timeout -k 10 200 python tools/phase_ticks.py fnav10 2>&1 | grep -v libdrm > $O/ticks_fnav10_span.txt; cat $O/ticks_fnav10_span.txt
FMARL_TICKS_SPAN=3 FMARL_LIB=$M timeout -k 10 200 python tools/phase_ticks.py fnav10 2>&1 | grep -v libdrm > $O/ticks_fnav10_span3.txt; cat $O/ticks_fnav10_span3.txt
timeout -k 10 200 python tools/ring_epb.py n10 0,25,0,25 span 2>&1 | grep -v libdrm > $O/n10_epb.txt; cat $O/n10_epb.txt
