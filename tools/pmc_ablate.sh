#!/bin/bash
# measurement aid: VALU / LDS instruction counts of the step kernel per phase (FMARL_ABLATE masks, -DFMARL_MEASURE build)
# usage: tools/pmc_ablate.sh [config] [mask ...]
CFG=${1:-cfg3}; shift
MASKS=${@:-0 1 2 4 8 16 32 31}
cd "$(dirname "$0")/../fair_marl_amd/csrc" && cp libfmarl.so libfmarl_ship.so && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -DFMARL_MEASURE -shared -fPIC -o libfmarl.so libfmarl.hip && cd ../..
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for m in $MASKS; do
  rm -rf /tmp/pa_$m
  FMARL_ABLATE=$m rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pa_$m -- python3 $ROOT/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --sync-reset > /tmp/pa_$m.log 2>&1
  python3 - <<PY
import csv, glob, collections, numpy as np
rows = list(csv.DictReader(open(glob.glob('/tmp/pa_$m/*/*counter_collection.csv')[0])))
d = collections.defaultdict(list)
for r in rows:
    if 'step_kernel' in r['Kernel_Name'] or 'formation_kernel<true>' in r['Kernel_Name'] or 'fairnav_kernel<true>' in r['Kernel_Name']:
        d[r['Counter_Name']].append(float(r['Counter_Value']))
print('ablate=%3s ' % '$m' + '  '.join('%s=%.4g' % (c, np.median(v)) for c, v in sorted(d.items())))
PY
done
cp $ROOT/fair_marl_amd/csrc/libfmarl_ship.so $ROOT/fair_marl_amd/csrc/libfmarl.so && rm $ROOT/fair_marl_amd/csrc/libfmarl_ship.so
