#!/bin/bash
# measurement aid (GPU box): vector instructions issued by a config's step kernel (one launch per step) under two library variants.
#   tools/pmc_valu_ab.sh <config> <variant> [<variant> ...]      variant "ship" = the shipped libfmarl.so (tools/mkvariant.sh builds the others)
#   COUNTERS="SQC_ICACHE_REQ SQC_ICACHE_MISSES ..." for another counter set (one pass: what the SQ block can count at once); LAUNCH=span for the span kernel
CFG=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  LIB=$GRAFT_REPO_ROOT/fair_marl_amd/csrc/variants/libfmarl_$v.so; [ "$v" = ship ] && LIB=$GRAFT_REPO_ROOT/fair_marl_amd/csrc/libfmarl.so
  OUT=$GRAFT_REPO_ROOT/gpurun_out/valu_${CFG}_$v; rm -rf $OUT; mkdir -p $OUT
  export FMARL_LIB=$LIB
  rocprofv3 --pmc ${COUNTERS:-SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS} --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --launch ${LAUNCH:-step} > $OUT/log.txt 2>&1
  python3 - <<PY
import csv, glob, collections, numpy as np
rows = list(csv.DictReader(open(glob.glob('$OUT/pmc/*/*counter_collection.csv')[0])))
d = collections.defaultdict(list)
for r in rows:
    k = r['Kernel_Name'].split('(')[0][:40]
    if 'step_kernel' in k or 'step_small' in k or 'span_kernel' in k or 'formation_kernel<true>' in k or 'fairnav_kernel<true' in k: d[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(d.items()): print('$v: %-36s %-22s n=%3d median=%.4g' % (k, c, len(v), np.median(v)))
PY
done
