#!/bin/bash
# measurement aid: L2 -> fabric request mix of a step kernel (are the stores full 64-byte writes, do they trigger reads?)
# usage (GPU box): tools/pmc_tcc.sh <config> [bench args]
CFG=${1:-n10}
OUT=$GRAFT_REPO_ROOT/gpurun_out/tcc_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_WRITE_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_STALL_sum TCC_WRITEBACK_sum"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-tune-placement ${@:2} > $OUT/log_$tag.txt 2>&1 || { tail -n 5 $OUT/log_$tag.txt; continue; }
done
python3 - <<PY
import csv, glob, collections, numpy as np
d = collections.defaultdict(list)
for f in glob.glob('$OUT/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        d[(r['Kernel_Name'].split('(')[0][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(d.items()):
    if 'step_kernel' in k or 'formation_kernel<true>' in k or 'fairnav_kernel<true>' in k: print('%-42s %-26s n=%3d median=%.4g' % (k, c, len(v), np.median(v)))
PY
