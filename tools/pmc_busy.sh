#!/bin/bash
# measurement aid (GPU box): how busy the vector ALUs are under a config's step kernel (one launch per step): SQ activity / wait counters.
#   tools/pmc_busy.sh <config>        SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES ~ the share of cycles a SIMD issues vector instructions
CFG=${1:-cfg4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/busy_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --launch step > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections, numpy as np
rows = list(csv.DictReader(open(glob.glob('$OUT/pmc/*/*counter_collection.csv')[0])))
d = collections.defaultdict(list)
for r in rows:
    k = r['Kernel_Name'].split('(')[0][:40]
    if 'step_kernel' in k or 'step_small' in k or 'formation_kernel<true>' in k or 'fairnav_kernel<true' in k: d[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(d.items()): print('%-42s %-22s n=%3d median=%.4g' % (k, c, len(v), np.median(v)))
PY
