#!/bin/bash
# measurement aid (GPU box): memory-side counters of the generic emission path (bench.py --config n10, the span kernel): the L2 -> fabric
# write requests (all / 64-byte / to DRAM / stalled), the L1 -> L2 write requests and all L2 requests -- two passes (three TCC counters
# per pass is what one pass takes).   tools/pmc_tcc_n10.sh [config=n10]
CFG=${1:-n10}
OUT=$GRAFT_REPO_ROOT/gpurun_out/tcc_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=1
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum" "TCP_TCC_WRITE_REQ_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum" "TCC_WRITE_sum TCC_WRITEBACK_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"; do
  rm -rf $OUT/p$P
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/p$P -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 25 --warmup 25 --no-cpu-baseline --no-secondary > $OUT/p$P.log 2>&1
  P=$((P+1))
done
python3 - <<PY
import csv, glob, collections, json
d = collections.defaultdict(list)
for f in glob.glob('$OUT/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'step_span' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
line = [l for l in open('$OUT/p1.log') if l.startswith('{')]
rl = json.loads(line[-1])['roofline'] if line else {}
v = {c: max(x) for c, x in d.items()}     # the longest span launch of the run (24 steps)
print('$CFG span kernel, the 24-step launch:', '  '.join('%s=%.4g' % (c, x) for c, x in sorted(v.items())))
if 'TCC_EA0_WRREQ_sum' in v:
    w, w64 = v['TCC_EA0_WRREQ_sum'], v.get('TCC_EA0_WRREQ_64B_sum', 0)
    print('write requests to the fabric: %.4g, %.1f %% of them 64-byte; bytes if the rest are 32-byte: %.4g; stalled cycles per request %.2f; L1->L2 write requests per fabric request %.2f'
          % (w, 100 * w64 / w, 64 * w64 + 32 * (w - w64), v.get('TCC_EA0_WRREQ_STALL_sum', 0) / w, v.get('TCP_TCC_WRITE_REQ_sum', 0) / w))
print('bench line: kernel %.4f ms per launch, %s steps per launch, frac %.3f' % (rl.get('kernel_avg_ms', 0), rl.get('kernel_steps_per_launch'), rl.get('frac', 0)))
PY
