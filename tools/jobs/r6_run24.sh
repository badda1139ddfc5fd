cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6y; mkdir -p $O
for rep in 1 2; do for v in ship fdirect; do L=fair_marl_amd/csrc/libfmarl.so; [ $v != ship ] && L=fair_marl_amd/csrc/variants/libfmarl_$v.so; for m in span eager; do echo -n "$v: " >> $O/ab_form_direct.txt; FMARL_LIB=$PWD/$L timeout -k 10 200 python tools/ring_epb.py cfg4 0 $m 2>&1 | grep -v libdrm >> $O/ab_form_direct.txt; done; done; done
cat $O/ab_form_direct.txt
