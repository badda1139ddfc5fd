cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6q; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -k "fnav2 or fnav10 or fnavw or store_pattern" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -8 $O/tests.log
