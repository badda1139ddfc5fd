#!/bin/bash
# measurement aid: step-kernel time against workgroups per CU (the -DFMARL_MEASURE build pads the dynamic LDS by FMARL_LDS_PAD bytes)
# usage: tools/occ_probe.sh [config] [pad ...]
CFG=${1:-cfg4}; shift
PADS=${@:-0 14000 28000}
cd "$(dirname "$0")/../fair_marl_amd/csrc" && cp libfmarl.so libfmarl_ship.so && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -DFMARL_MEASURE -shared -fPIC -o libfmarl.so libfmarl.hip && cd ../..
for pad in $PADS; do
  FMARL_LDS_PAD=$pad python bench.py --config $CFG --steps 100 --warmup 25 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lds_pad=%6s  ms_per_step=%.3f  kernel_avg_ms=%.3f' % ('$pad', d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done
cp fair_marl_amd/csrc/libfmarl_ship.so fair_marl_amd/csrc/libfmarl.so && rm fair_marl_amd/csrc/libfmarl_ship.so
