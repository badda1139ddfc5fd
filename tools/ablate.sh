#!/bin/bash
# measurement aid (GPU box): step time with phases of a step kernel skipped (FMARL_ABLATE bit mask; the bits are the
# FMARL_SKIP(p, bit) sites of the kernel).  usage: tools/ablate.sh [config] [mask ...]
# Needs the -DFMARL_MEASURE variant built in the container first: tools/mkvariant.sh measure -DFMARL_MEASURE
# (the shipped build has no such switch and is never touched).
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
CFG=${1:-cfg3}; shift
MASKS=${@:-0 1 2 4 8 16 31 32 33 35 39 47 63}
for m in $MASKS; do
  FMARL_LIB=$R/fair_marl_amd/csrc/variants/libfmarl_measure.so FMARL_ABLATE=$m python bench.py --config $CFG --steps 100 --warmup 25 --no-cpu-baseline --no-secondary $ABL_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=%3s  ms_per_step=%.3f  kernel_avg_ms=%.3f' % ('$m', d['ms_per_step'], d['roofline']['kernel_avg_ms']))"
done
