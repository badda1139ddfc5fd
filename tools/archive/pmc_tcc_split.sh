#!/bin/bash
# measurement aid: L2 -> fabric write requests of a step kernel with node_obs / adj emission switched off one at a time
# usage (GPU box): tools/pmc_tcc_split.sh <config>
CFG=${1:-n10}
OUT=$GRAFT_REPO_ROOT/gpurun_out/tccs_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3; do
  export EMIT_SPLIT_ONLY=$v
  rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/v$v -- python3 $GRAFT_REPO_ROOT/tools/emit_split.py $CFG 30 > $OUT/log_v$v.txt 2>&1 || { tail -n 5 $OUT/log_v$v.txt; continue; }
  grep ms_per_step $OUT/log_v$v.txt | cut -c1-60
  python3 - <<PY
import csv, glob, collections, numpy as np
d = collections.defaultdict(list)
for f in glob.glob('$OUT/v$v/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        d[(r['Kernel_Name'].split('(')[0][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(d.items()):
    if 'step_kernel' in k or 'formation_kernel<true>' in k or 'fairnav_kernel<true>' in k: print('   %-30s %-24s median=%.4g' % (k, c, np.median(v)))
PY
done
