cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6w; mkdir -p $O
for rep in 1 2; do for c in fnav4 fnav5; do timeout -k 10 200 python tools/fnav_lines.py $c eager,span 1 2>&1 | grep -v libdrm >> $O/fnav45.txt; done; done
cat $O/fnav45.txt
