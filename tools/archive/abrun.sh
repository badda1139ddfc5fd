#!/bin/bash
# measurement aid (GPU box): alternate library variants on ONE box (box-to-box spread is larger than most changes).
#   tools/archive/abrun.sh <config> <steps> <rounds> <name> [<name> ...]      name "ship" = the shipped libfmarl.so
# Extra bench.py arguments through ABRUN_ARGS; environment for the variants (e.g. FMARL_ABLATE) is inherited.
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
CFG=$1; STEPS=$2; ROUNDS=$3; shift 3
for r in $(seq 1 $ROUNDS); do for v in "$@"; do
  LIB=$R/fair_marl_amd/csrc/variants/libfmarl_$v.so; [ "$v" = ship ] && LIB=$R/fair_marl_amd/csrc/libfmarl.so
  echo -n "$v: "
  FMARL_LIB=$LIB python bench.py --config $CFG --steps $STEPS --warmup 25 --no-cpu-baseline --no-secondary $ABRUN_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms_per_step=%.4f  kernel_avg_ms=%.4f  frac=%.3f  ceiling=%s' % (d['ms_per_step'], r['kernel_avg_ms'], r['frac'], r.get('store_ceiling_ms')))"
done; done
