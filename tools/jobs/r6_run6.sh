set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "fairnav or fnav or random_small or full_size" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -5 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
M=$PWD/fair_marl_amd/csrc/variants/libfmarl_measure.so
FMARL_LIB=$M timeout -k 10 200 python tools/phase_ticks.py fnav 0.5 600 2>&1 | grep -v libdrm > $O/ticks_fnav_steady.txt; cat $O/ticks_fnav_steady.txt
for rep in 1 2; do timeout -k 10 300 python tools/fnav_lines.py fnav eager,span,steady,steady-span 1 2>&1 | grep -v libdrm >> $O/ab_fnav.txt; done
cat $O/ab_fnav.txt
timeout -k 10 200 python tools/fnav_lines.py fnav10 eager,span,span5,span2 1 2>&1 | grep -v libdrm > $O/fnav10_spans.txt; cat $O/fnav10_spans.txt
export COUNTERS="SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
LAUNCH=step timeout -k 10 300 bash tools/pmc_valu_ab.sh fnav10 ship > $O/icache_fnav10_step.txt 2>&1; cat $O/icache_fnav10_step.txt
LAUNCH=span timeout -k 10 300 bash tools/pmc_valu_ab.sh fnav10 ship > $O/icache_fnav10_span.txt 2>&1; cat $O/icache_fnav10_span.txt
