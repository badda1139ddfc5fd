cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6aa; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "full_size or random_small or shape_instances or traj or runner" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
for rep in 1 2; do for v in ship prev; do L=fair_marl_amd/csrc/libfmarl.so; [ $v != ship ] && L=fair_marl_amd/csrc/variants/libfmarl_$v.so; for cm in "n10 span" "n10 eager" "cfg2 span" "fnav span" "fnav eager" "cfg4 span"; do set -- $cm; echo -n "$v: " >> $O/ab_adj4.txt; FMARL_LIB=$PWD/$L timeout -k 10 200 python tools/ring_epb.py $1 0 $2 2>&1 | grep -v libdrm >> $O/ab_adj4.txt; done; done; done
cat $O/ab_adj4.txt
