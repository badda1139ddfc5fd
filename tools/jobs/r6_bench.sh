set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
S=$(date +%s); timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_driver_detail.json > $O/bench_driver.json 2> $O/bench_driver.err; echo "rc=$? bytes=$(wc -c < $O/bench_driver.json) wall=$(( $(date +%s) - S )) s"
timeout -k 10 400 python3 bench.py --detail $O/bench_default_detail.json > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$? bytes=$(wc -c < $O/bench_default.json)"
cat $O/bench_driver.json
