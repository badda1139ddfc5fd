#!/bin/bash
# measurement aid (GPU box): one launch per step into ONE output set (the engine's own node_obs / adj) -- plain allocations
# (FMARL_RING_SPREAD=0) against arrays made of hipMemCreate pieces (the default).  usage: bash tools/archive/oneset_ab.sh [config=cfg3]
cd "$(dirname "$0")/.."
for r in 1 2; do for sp in 0 1; do
  echo -n "pieces=$sp: "; FMARL_RING_SPREAD=$sp python bench.py --config ${1:-cfg3} --launch step --slots same --steps 100 --warmup 25 --no-cpu-baseline --no-secondary 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f kernel %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['roofline']['frac']))"
done; done
