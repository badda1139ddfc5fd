set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6o; mkdir -p $O
for rep in 1 2 3; do for v in ship fs3; do L=fair_marl_amd/csrc/libfmarl.so; [ $v != ship ] && L=fair_marl_amd/csrc/variants/libfmarl_$v.so; echo -n "$v: " >> $O/ab_form_blocks.txt; FMARL_LIB=$PWD/$L timeout -k 10 200 python tools/ring_epb.py cfg4 0 span 2>&1 | grep -v libdrm >> $O/ab_form_blocks.txt; done; done
cat $O/ab_form_blocks.txt
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')"
