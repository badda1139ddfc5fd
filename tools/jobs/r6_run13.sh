set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "shape_instances or falls_back or fairnav or fnav" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -5 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
for rep in 1 2; do for g in 0 1; do
  echo "== FMARL_GENERIC_SHAPES=$g" >> $O/shape_ab.txt
  FMARL_GENERIC_SHAPES=$g timeout -k 10 200 python tools/fnav_lines.py fnav10 eager 1 2>&1 | grep -v libdrm >> $O/shape_ab.txt
done; done
cat $O/shape_ab.txt
