#!/bin/bash
# measurement aid: per-channel TCC write counters of the fastest and the slowest (node_obs, adj) allocation pair -- only worth
# collecting on a box where the pairs differ at all (on many boxes they do not): times the pairs first, profiles if the spread >= 5 %.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/placement_tcc.py > $R/gpurun_out/placement_hunt_plain.log 2>&1
SPREAD=$(python3 - <<PY
import re
l=[x for x in open('$R/gpurun_out/placement_hunt_plain.log') if x.startswith('PAIR_TIMES')][0]
v=[float(x.split('=')[1]) for x in l.split()[1:]]
print('%.3f' % (max(v)/min(v)-1))
PY
)
echo "spread of the 18 pairs on this box: $SPREAD"
grep LAST_TEN $R/gpurun_out/placement_hunt_plain.log
python3 -c "import sys; sys.exit(0 if float('$SPREAD') >= 0.05 else 1)" || { echo "no placement effect on this box: nothing to profile"; exit 0; }
for y in tcc_minmax tcc_xcc; do
  C=$(grep -o "name: FM_[A-Z0-9_]*" $R/tools/pmc/$y.yaml | sed "s/name: //" | tr "\n" " ")
  rocprofv3 -E $R/tools/pmc/$y.yaml --pmc $C --output-format csv -d /tmp/ph_$y -- python3 $R/tools/placement_tcc.py > $R/gpurun_out/placement_hunt_$y.log 2>&1
  python3 $R/tools/placement_tcc.py --summarize /tmp/ph_$y $R/gpurun_out/placement_hunt_$y.log "$y" > $R/gpurun_out/r2_placement_hunt_$y.md 2>&1
  cat $R/gpurun_out/r2_placement_hunt_$y.md
done
rocprofv3 --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_sum --output-format csv -d /tmp/ph_sum -- python3 $R/tools/placement_tcc.py > $R/gpurun_out/placement_hunt_sum.log 2>&1
python3 $R/tools/placement_tcc.py --summarize /tmp/ph_sum $R/gpurun_out/placement_hunt_sum.log "sums" > $R/gpurun_out/r2_placement_hunt_sum.md 2>&1
cat $R/gpurun_out/r2_placement_hunt_sum.md
