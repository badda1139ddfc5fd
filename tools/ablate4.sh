#!/bin/bash
# measurement aid: step time with phases of the step kernel skipped (FMARL_ABLATE bit mask).
# Runs on the GPU box with a -DFMARL_MEASURE build of the library (the shipped build has no such switch); restores it after.
cd "$(dirname "$0")/../fair_marl_amd/csrc" && cp libfmarl.so libfmarl_ship.so && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -DFMARL_MEASURE -shared -fPIC -o libfmarl.so libfmarl.hip && cd ../..
for m in 0 1 32 64 128 224 225; do
  FMARL_ABLATE=$m python bench.py --config cfg4 --steps 50 --warmup 25 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=%3s  ms_per_step=%.3f  kernel_avg_ms=%.3f' % ('$m', d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done
cp fair_marl_amd/csrc/libfmarl_ship.so fair_marl_amd/csrc/libfmarl.so && rm fair_marl_amd/csrc/libfmarl_ship.so
