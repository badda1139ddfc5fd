#!/bin/bash
# Run on the GPU box (via gpurun): bench line, rocprofv3 kernel-trace stats and the two HBM-traffic
# PMC passes for the same bench command.  Raw output under gpurun_out/<tag>/, summarised by
# tools/summarize_profile.py into profiles/.
TAG=${1:-r1}
CFG=${2:-cfg3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --config $CFG --no-secondary --detail $OUT/bench_detail.json $PROF_ARGS > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 200 --warmup 50 --no-cpu-baseline --no-secondary $PROF_ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 50 --warmup 25 --no-cpu-baseline --no-secondary $PROF_ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 50 --warmup 25 --no-cpu-baseline --no-secondary $PROF_ARGS > $OUT/pmc_write.log 2>&1
# the driver's own command (python3 bench.py --gpus 1 --steps 20 --warmup 5) under the kernel trace: its span launch covers 19 steps
if [ "$CFG" = cfg3 ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/trace_driver.log 2>&1
fi
cat $OUT/bench.json
find $OUT -name '*.csv' | head -20
