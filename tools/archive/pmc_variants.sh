#!/bin/bash
# measurement aid: HBM counters of the step kernel for prebuilt library variants (fair_marl_amd/csrc/libfmarl_<tag>.so), one box
ROOT=$GRAFT_REPO_ROOT; CFG=${CFG:-cfg3}
cd $ROOT/fair_marl_amd/csrc && cp libfmarl.so libfmarl_base.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp $ROOT/fair_marl_amd/csrc/libfmarl_$v.so $ROOT/fair_marl_amd/csrc/libfmarl.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pv; rocprofv3 --pmc $c --output-format csv -d /tmp/pv -- python3 $ROOT/bench.py --config $CFG --steps 30 --warmup 25 --no-cpu-baseline > /dev/null 2>&1
    python3 - "$v" "$c" <<'PY'
import csv, glob, sys
rows = [float(r['Counter_Value']) for f in glob.glob('/tmp/pv/*/*counter_collection.csv') for r in csv.DictReader(open(f))
        if 'step_kernel' in r['Kernel_Name'] and r['Counter_Name'] == sys.argv[2]]
print('%-8s %-10s mean %.0f KiB max %.0f KiB over %d launches' % (sys.argv[1], sys.argv[2], sum(rows) / len(rows), max(rows), len(rows)))
PY
  done
done
cp $ROOT/fair_marl_amd/csrc/libfmarl_base.so $ROOT/fair_marl_amd/csrc/libfmarl.so
