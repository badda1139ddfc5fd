#!/bin/bash
# measurement aid: step-kernel time against envs per workgroup (FMARL_EPB) and LDS padding (FMARL_LDS_PAD: fewer resident
# workgroups per CU) in a -DFMARL_MEASURE build.  usage: tools/epb_probe.sh <config> "<epb>:<pad> ..."
CFG=${1:-n10}; shift
COMBOS=${@:-"0:0"}
cd "$(dirname "$0")/../fair_marl_amd/csrc" && cp libfmarl.so libfmarl_ship.so && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -DFMARL_MEASURE -shared -fPIC -o libfmarl.so libfmarl.hip && cd ../..
for c in $COMBOS; do
  epb=${c%%:*}; pad=${c##*:}
  FMARL_EPB=$epb FMARL_LDS_PAD=$pad python bench.py --config $CFG --steps 200 --warmup 25 --no-cpu-baseline $EPB_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('epb=%3s lds_pad=%6s  ms_per_step=%.3f  kernel_avg_ms=%.3f' % ('$epb', '$pad', d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done
cp fair_marl_amd/csrc/libfmarl_ship.so fair_marl_amd/csrc/libfmarl.so && rm fair_marl_amd/csrc/libfmarl_ship.so
