cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6ab; mkdir -p $O
for rep in 1 2; do for v in ship prev; do L=fair_marl_amd/csrc/libfmarl.so; [ $v != ship ] && L=fair_marl_amd/csrc/variants/libfmarl_$v.so; for cm in "n10 span" "cfg2 span" "cfg2 eager" "fnav10 eager"; do set -- $cm; echo -n "$v: " >> $O/ab_adj4.txt; if [ $1 = fnav10 ]; then FMARL_LIB=$PWD/$L timeout -k 10 200 python tools/fnav_lines.py fnav10 eager 1 2>&1 | grep -v libdrm >> $O/ab_adj4.txt; else FMARL_LIB=$PWD/$L timeout -k 10 200 python tools/ring_epb.py $1 0 $2 2>&1 | grep -v libdrm >> $O/ab_adj4.txt; fi; done; done; done
cat $O/ab_adj4.txt
