#!/bin/bash
# measurement aid (GPU box): VALU / LDS instruction counts of the step kernel per phase (FMARL_ABLATE masks of the
# -DFMARL_MEASURE variant: build it first in the container with tools/mkvariant.sh measure -DFMARL_MEASURE; the shipped
# library is never touched).  usage: tools/pmc_ablate.sh [config] [mask ...]
R=$(cd "$(dirname "$0")/.." && pwd)
CFG=${1:-cfg3}; shift
MASKS=${@:-0 1 2 4 8 16 32 31}
cd /tmp && export TMPDIR=/tmp
for m in $MASKS; do
  rm -rf /tmp/pa_$m
  FMARL_LIB=$R/fair_marl_amd/csrc/variants/libfmarl_measure.so FMARL_ABLATE=$m rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pa_$m -- python3 $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --eager --sync-reset > /tmp/pa_$m.log 2>&1
  python3 - <<PY
import csv, glob, collections, numpy as np
rows = list(csv.DictReader(open(glob.glob('/tmp/pa_$m/*/*counter_collection.csv')[0])))
d = collections.defaultdict(list)
for r in rows:
    if 'step_kernel' in r['Kernel_Name'] or 'formation_kernel<true>' in r['Kernel_Name'] or 'fairnav_kernel<true>' in r['Kernel_Name']:
        d[r['Counter_Name']].append(float(r['Counter_Value']))
print('ablate=%3s ' % '$m' + '  '.join('%s=%.4g' % (c, np.median(v)) for c, v in sorted(d.items())))
PY
done
