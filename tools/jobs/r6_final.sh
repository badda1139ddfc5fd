cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6z; mkdir -p $O
S=$(date +%s); timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$? wall=$(( $(date +%s) - S )) s" >> $O/tests.log; tail -4 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S=$(date +%s); timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_driver_detail.json > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$? bytes=$(wc -c < $O/bench_driver.json) wall=$(( $(date +%s) - S )) s"
