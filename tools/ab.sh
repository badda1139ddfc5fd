#!/bin/bash
# measurement aid: A/B of the working tree's library against the library built from a git revision, alternating on ONE box
# (box-to-box spread is larger than most changes).  The revision must have the same C-ABI as the working tree.
# build here:   tools/ab.sh build <rev>        (writes fair_marl_amd/csrc/libfmarl_ref.so)
# on the box:   tools/ab.sh run <config> [steps]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  rm -rf /tmp/ab_src && mkdir -p /tmp/ab_src && (cd $R && git archive $2 fair_marl_amd/csrc include | tar -x -C /tmp/ab_src)
  (cd /tmp/ab_src/fair_marl_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -shared -fPIC -o $R/fair_marl_amd/csrc/libfmarl_ref.so libfmarl.hip)
  echo "built libfmarl_ref.so from $2"
else
  cd $R/fair_marl_amd/csrc; cp libfmarl.so libfmarl_new.so
  for r in 1 2 3; do for v in ref new; do cp libfmarl_$v.so libfmarl.so; echo -n "$v: "; (cd $R; ./tools/quick.sh ${3:-400} --config $2); done; done
  cp libfmarl_new.so libfmarl.so; rm libfmarl_new.so
fi
