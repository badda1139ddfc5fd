set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "pipelined or time_slot or bench_ or fairnav or ring or random_small" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -3 $O/tests.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$? bytes=$(wc -c < $O/bench_driver.json)"
cp bench_detail.json $O/bench_detail.json
for v in ship r5 ship r5; do L=fair_marl_amd/csrc/libfmarl.so; [ $v = r5 ] && L=fair_marl_amd/csrc/variants/libfmarl_r5.so; echo "== $v" >> $O/ab_fnav10.txt; FMARL_LIB=$PWD/$L timeout -k 10 200 python tools/fnav_lines.py fnav10 eager,span 1 >> $O/ab_fnav10.txt 2>&1; done
tail -12 $O/ab_fnav10.txt
timeout -k 10 200 tools/vmm_fault_repro 8 > $O/vmm_fault_repro.txt 2>&1; echo "repro rc=$?"; grep SUMMARY $O/vmm_fault_repro.txt
