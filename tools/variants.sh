#!/bin/bash
# measurement aid: compare prebuilt library variants (fair_marl_amd/csrc/libfmarl_<tag>.so) on one box
cd fair_marl_amd/csrc; cp libfmarl.so libfmarl_base.so
for v in "$@"; do cp libfmarl_$v.so libfmarl.so; echo "variant $v"; (cd ../..; ./tools/quick.sh 300 $CFG); done
cp libfmarl_base.so libfmarl.so
