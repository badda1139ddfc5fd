#!/bin/bash
# measurement aid (build container): build a variant of the library next to the shipped one, never over it.
#   tools/mkvariant.sh <name> [extra hipcc flags ...]     -> fair_marl_amd/csrc/variants/libfmarl_<name>.so  (working tree)
#   REV=<git rev> tools/mkvariant.sh <name> [flags ...]    -> the same from a revision's sources (must have today's C-ABI)
# Variants are git-ignored (*.so) but travel to the GPU box; select one with FMARL_LIB=<path> (fair_marl_amd/_lib.py).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
SRC=$R
if [ -n "$REV" ]; then
  SRC=/tmp/variant_src_$NAME; rm -rf $SRC; mkdir -p $SRC
  (cd $R && git archive $REV fair_marl_amd/csrc include | tar -x -C $SRC)
fi
mkdir -p $R/fair_marl_amd/csrc/variants
(cd $SRC/fair_marl_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm \
   -shared -fPIC "$@" -o $R/fair_marl_amd/csrc/variants/libfmarl_$NAME.so libfmarl.hip)
echo "built fair_marl_amd/csrc/variants/libfmarl_$NAME.so from ${REV:-the working tree} $*"
