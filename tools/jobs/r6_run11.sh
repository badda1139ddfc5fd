set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6k; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -5 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
for rep in 1 2; do for g in 0 1; do
  echo "== FMARL_GENERIC_SHAPES=$g" >> $O/shape_ab.txt
  for cm in "cfg2 span" "cfg2 eager" "n10 span" "n10 eager" "cfg3 eager" "fnav span" "fnav eager"; do set -- $cm; FMARL_GENERIC_SHAPES=$g timeout -k 10 200 python tools/ring_epb.py $1 0 $2 2>&1 | grep -v libdrm >> $O/shape_ab.txt; done
done; done
cat $O/shape_ab.txt
