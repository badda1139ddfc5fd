set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6b; mkdir -p $O
for rep in 1 2; do for v in ship nc3 fc3 r5; do L=fair_marl_amd/csrc/libfmarl.so; [ $v != ship ] && L=fair_marl_amd/csrc/variants/libfmarl_$v.so; echo "== $v" >> $O/ab_fnav10.txt; FMARL_LIB=$PWD/$L timeout -k 10 200 python tools/fnav_lines.py fnav10 eager,span 1 2>&1 | grep -v libdrm >> $O/ab_fnav10.txt; done; done
cat $O/ab_fnav10.txt
timeout -k 10 300 python tools/n10_pattern.py n10 > $O/n10_pattern.txt 2>&1; cat $O/n10_pattern.txt
timeout -k 10 900 bash tools/pmc_tcc_n10.sh n10 > $O/n10_tcc.txt 2>&1; cat $O/n10_tcc.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -5 $O/tests.log
