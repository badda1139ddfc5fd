#!/bin/bash
# measurement aid: step time with phases of step_kernel skipped (FMARL_ABLATE bit mask)
for m in 0 1 2 4 8 16 32 33 35 39 47 63; do
  FMARL_ABLATE=$m python bench.py --steps 50 --warmup 25 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=%3s  ms_per_step=%.3f  kernel_avg_ms=%.3f' % ('$m', d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done
