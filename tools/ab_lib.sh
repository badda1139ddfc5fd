#!/bin/bash
# measurement aid (GPU box): tools/ring_epb.py lines of several library variants, alternating, on ONE box.
#   tools/ab_lib.sh <config> <epb> <mode> <rounds> <name> [<name> ...]     name "ship" = the shipped libfmarl.so
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
CFG=$1; EPB=$2; MODE=$3; ROUNDS=$4; shift 4
for r in $(seq 1 $ROUNDS); do for v in "$@"; do
  LIB=$R/fair_marl_amd/csrc/variants/libfmarl_$v.so; [ "$v" = ship ] && LIB=$R/fair_marl_amd/csrc/libfmarl.so
  echo -n "$v: "; FMARL_LIB=$LIB python tools/ring_epb.py $CFG $EPB $MODE 2>&1 | grep -v libdrm
done; done
