#!/bin/bash
# measurement aid: where a step kernel's wave-cycles go (MI355X_MICROARCH.md "rocprofv3 PMC slots": WAIT_ANY + WAIT_INST_ANY +
# ACTIVE_INST_ANY ~ WAVE_CYCLES, all in quad-cycles), plus the VALU / LDS / SALU / VMEM active shares
CFG=${1:-cfg4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/wait_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-secondary ${@:2} > $OUT/log.txt 2>&1
tail -n 3 $OUT/log.txt | cut -c1-200
python3 - <<PY
import csv, glob, collections, numpy as np
rows = list(csv.DictReader(open(glob.glob('$OUT/pmc/*/*counter_collection.csv')[0])))
d = collections.defaultdict(list)
for r in rows:
    d[(r['Kernel_Name'].split('(')[0][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(d.items()):
    if 'step_kernel' in k or 'formation_kernel<true>' in k or 'fairnav_kernel<true>' in k: print('%-42s %-22s n=%3d median=%.4g' % (k, c, len(v), np.median(v)))
PY
