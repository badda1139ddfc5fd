cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6r; mkdir -p $O
for us in 0 20 40 80 0 40 160; do echo -n "stagger $us us: " >> $O/stagger_cfg3.txt; FMARL_STAGGER_US=$us timeout -k 10 300 python tools/ring_epb.py cfg3 0 eager 2>&1 | grep -v libdrm >> $O/stagger_cfg3.txt; done
cat $O/stagger_cfg3.txt
for us in 0 10 20 40 0 20; do echo -n "stagger $us us: " >> $O/stagger_n10.txt; FMARL_STAGGER_US=$us timeout -k 10 300 python tools/ring_epb.py n10 0 eager 2>&1 | grep -v libdrm >> $O/stagger_n10.txt; done
cat $O/stagger_n10.txt
