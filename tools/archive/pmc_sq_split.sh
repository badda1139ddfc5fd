#!/bin/bash
# measurement aid: SQ counters of a step kernel with node_obs / adj emission switched off one at a time
# usage (GPU box): tools/pmc_sq_split.sh <config> [variants, default "0 1 2 3"]
CFG=${1:-n10}
VARS=${2:-0 1 2 3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/sqs_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  export EMIT_SPLIT_ONLY=$v
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/v${v}_$i -- python3 $GRAFT_REPO_ROOT/tools/emit_split.py $CFG 30 > $OUT/log_v${v}_$i.txt 2>&1 || { tail -n 5 $OUT/log_v${v}_$i.txt; continue; }
  done
  grep ms_per_step $OUT/log_v${v}_1.txt | cut -c1-60
  python3 - <<PY
import csv, glob, collections, numpy as np
d = collections.defaultdict(list)
for f in glob.glob('$OUT/v${v}_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        d[(r['Kernel_Name'].split('(')[0][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(d.items()):
    if 'step_kernel' in k or 'formation_kernel<true>' in k or 'fairnav_kernel<true>' in k: print('   %-30s %-24s median=%.4g' % (k, c, np.median(v)))
PY
done
