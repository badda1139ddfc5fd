set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6c; mkdir -p $O
for rep in 1 2; do for v in ship nc3 fc3; do L=fair_marl_amd/csrc/libfmarl.so; [ $v != ship ] && L=fair_marl_amd/csrc/variants/libfmarl_$v.so; echo "== $v" >> $O/ab_fnav10.txt; FMARL_LIB=$PWD/$L timeout -k 10 200 python tools/fnav_lines.py fnav10 eager,span 1 2>&1 | grep -v libdrm >> $O/ab_fnav10.txt; done; done
cat $O/ab_fnav10.txt
timeout -k 10 200 python tools/ring_epb.py n10 25,24,25,24,16 span 2>&1 | grep -v libdrm > $O/n10_epb.txt; cat $O/n10_epb.txt
timeout -k 10 200 python tools/ring_epb.py n10 25,24,25,24 eager 2>&1 | grep -v libdrm >> $O/n10_epb.txt; tail -4 $O/n10_epb.txt
timeout -k 10 120 python tools/jobs/r6_oom_probe.py 2>&1 | grep -v libdrm > $O/oom_probe.txt; cat $O/oom_probe.txt
