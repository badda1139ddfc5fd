#!/bin/bash
# measurement aid (GPU box): LDS bank conflicts and instruction mix of a config's step kernel (one launch per step).
#   tools/pmc_lds.sh <config>
CFG=${1:-n10}
OUT=$GRAFT_REPO_ROOT/gpurun_out/lds_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --launch step > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections, numpy as np
rows = list(csv.DictReader(open(glob.glob('$OUT/pmc/*/*counter_collection.csv')[0])))
d = collections.defaultdict(list)
for r in rows:
    k = r['Kernel_Name'].split('(')[0][:40]
    if 'step_kernel' in k or 'formation_kernel<true>' in k or 'fairnav_kernel<true>' in k: d[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(d.items()): print('%-42s %-22s n=%3d median=%.4g' % (k, c, len(v), np.median(v)))
PY
