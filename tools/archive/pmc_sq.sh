#!/bin/bash
# measurement aid: SQ counters of the step kernel (one --pmc pass, 8 SQ slots; MI355X_MICROARCH.md "rocprofv3 PMC slots")
CFG=${1:-cfg3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections, numpy as np
rows = list(csv.DictReader(open(glob.glob('$OUT/pmc/*/*counter_collection.csv')[0])))
d = collections.defaultdict(list)
for r in rows:
    d[(r['Kernel_Name'].split('(')[0][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(d.items()):
    if 'fmarl' in k: print('%-42s %-22s n=%3d mean=%.4g max=%.4g' % (k, c, len(v), np.mean(v), max(v)))
PY
