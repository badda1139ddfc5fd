cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6ac; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "fairnav or fnav or random_small or shape_instances" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
for rep in 1 2; do for v in ship prev; do L=fair_marl_amd/csrc/libfmarl.so; [ $v != ship ] && L=fair_marl_amd/csrc/variants/libfmarl_$v.so; echo "== $v" >> $O/ab_walk.txt; FMARL_LIB=$PWD/$L timeout -k 10 300 python tools/fnav_lines.py fnav10 eager 1 2>&1 | grep -v libdrm >> $O/ab_walk.txt; FMARL_LIB=$PWD/$L timeout -k 10 300 python tools/fnav_lines.py fnav eager,span,steady,steady-span 1 2>&1 | grep -v libdrm >> $O/ab_walk.txt; FMARL_LIB=$PWD/$L timeout -k 10 300 python tools/fnav_lines.py fnav6 eager 1 2>&1 | grep -v libdrm >> $O/ab_walk.txt; done; done
cat $O/ab_walk.txt
