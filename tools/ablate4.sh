#!/bin/bash
# measurement aid: formation kernel time with phases skipped (FMARL_ABLATE: 1 pair forces, 32 emission, 64 matchings, 128 occupancy walk)
for m in 0 1 32 64 128 224 225; do
  FMARL_ABLATE=$m python bench.py --config cfg4 --steps 50 --warmup 25 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=%3s  ms_per_step=%.3f  kernel_avg_ms=%.3f' % ('$m', d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done
