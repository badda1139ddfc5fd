#!/bin/bash
# measurement aid (GPU box): whole revisions against the working tree on ONE box -- each revision is a git worktree under _ab/<rev>
# with its own library and its own bench.py (build container: git worktree add _ab/<rev> <rev>; build it there).
#   tools/archive/ab_rev.sh <config> <steps> <rounds> <rev> [<rev> ...]     "." = the working tree
R=$(cd "$(dirname "$0")/.." && pwd)
CFG=$1; STEPS=$2; ROUNDS=$3; shift 3
for r in $(seq 1 $ROUNDS); do for v in "$@"; do
  D=$R/_ab/$v; [ "$v" = . ] && D=$R
  EXTRA=""; grep -q no-secondary $D/bench.py && EXTRA="--no-secondary"
  echo -n "$v: "
  (cd $D && python bench.py --config $CFG --steps $STEPS --warmup 25 --no-cpu-baseline $EXTRA 2>/dev/null) | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms_per_step=%.4f  kernel_avg_ms=%.4f  frac=%.3f' % (d['ms_per_step'], r['kernel_avg_ms'], r['frac']))"
done; done
