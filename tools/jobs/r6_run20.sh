cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6u; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "fairnav or fnav or random_small or lexifair or shape_instances" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
for rep in 1 2; do for v in ship prev; do L=fair_marl_amd/csrc/libfmarl.so; [ $v != ship ] && L=fair_marl_amd/csrc/variants/libfmarl_$v.so; for c in fnav10 fnav6; do echo -n "$v: " >> $O/ab_warm.txt; FMARL_LIB=$PWD/$L timeout -k 10 300 python tools/fnav_lines.py $c eager 1 2>&1 | grep -v libdrm >> $O/ab_warm.txt; done; done; done
cat $O/ab_warm.txt
