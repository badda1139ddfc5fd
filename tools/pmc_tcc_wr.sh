#!/bin/bash
# measurement aid (GPU box): the L2 -> memory-side write requests of the span kernel, with a time slot per step (--slots ring) and with ONE
# rewritten output set (--slots same): all write requests, those addressed to DRAM, and the stalls behind them (verdict round 3, item 1d).
# (three TCC counters per pass: five exceeded what one pass can collect and the profiler aborted)
#   tools/pmc_tcc_wr.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/tccwr
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for s in ring same; do
  rm -rf $OUT/$s
  timeout -k 10 240 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum --output-format csv -d $OUT/$s -- python3 $GRAFT_REPO_ROOT/bench.py --slots $s --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/$s.log 2>&1
  python3 - <<PY
import csv, glob, collections, json
rows = list(csv.DictReader(open(glob.glob('$OUT/$s/*/*counter_collection.csv')[0])))
d = collections.defaultdict(list)
for r in rows:
    if 'step_span_kernel' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
line = [l for l in open('$OUT/$s.log') if l.startswith('{')]
rl = json.loads(line[-1])['roofline'] if line else {}
print('--slots $s: span launches %s; the 19-step launch: %s; bench line: kernel %.3f ms per launch, frac %.3f' % (len(next(iter(d.values()), [])), '  '.join('%s=%.4g' % (c, v[-1]) for c, v in sorted(d.items())), rl.get('kernel_avg_ms', 0), rl.get('frac', 0)))
PY
done
