set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6h; mkdir -p $O
for c in fnav4 fnav5 fnav6 fnav8; do for v in ship fc4 fc3; do L=fair_marl_amd/csrc/libfmarl.so; [ $v != ship ] && L=fair_marl_amd/csrc/variants/libfmarl_$v.so; echo "== $c $v" >> $O/ab_fnavN.txt; FMARL_LIB=$PWD/$L timeout -k 10 200 python tools/fnav_lines.py $c eager,span 1 2>&1 | grep -v libdrm >> $O/ab_fnavN.txt; done; done
cat $O/ab_fnavN.txt
