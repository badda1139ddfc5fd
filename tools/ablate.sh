#!/bin/bash
# measurement aid: step time with phases of a step kernel skipped (FMARL_ABLATE bit mask; the bits are the
# FMARL_SKIP(p, bit) sites of the kernel).  usage: tools/ablate.sh [config] [mask ...]
# Runs on the GPU box with a -DFMARL_MEASURE build of the library (the shipped build has no such switch); restores it after.
CFG=${1:-cfg3}; shift
MASKS=${@:-0 1 2 4 8 16 31 32 33 35 39 47 63}
cd "$(dirname "$0")/../fair_marl_amd/csrc" && cp libfmarl.so libfmarl_ship.so && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -DFMARL_MEASURE -shared -fPIC -o libfmarl.so libfmarl.hip && cd ../..
for m in $MASKS; do
  FMARL_ABLATE=$m python bench.py --config $CFG --steps 100 --warmup 25 --no-cpu-baseline $ABL_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=%3s  ms_per_step=%.3f  kernel_avg_ms=%.3f' % ('$m', d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done
cp fair_marl_amd/csrc/libfmarl_ship.so fair_marl_amd/csrc/libfmarl.so && rm fair_marl_amd/csrc/libfmarl_ship.so
