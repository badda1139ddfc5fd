#!/bin/bash
# measurement aid: rocprofv3 per-kernel averages of a short bench run
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $GRAFT_REPO_ROOT/bench.py --steps ${1:-100} --warmup 25 --no-cpu-baseline ${@:2} > /tmp/ks.log 2>&1
python3 - <<'PY'
import csv, glob, json
line = [l for l in open('/tmp/ks.log') if l.startswith('{')][-1]
d = json.loads(line); print('ms_per_step=%.3f kernel_avg_ms=%.3f' % (d['ms_per_step'], d['roofline']['kernel_avg_ms']))
for r in list(csv.DictReader(open(glob.glob('/tmp/ks/*/*kernel_stats.csv')[0])))[:6]:
    print('%-44s calls=%4s avg_us=%9.1f min_us=%9.1f' % (r['Name'][:44], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
