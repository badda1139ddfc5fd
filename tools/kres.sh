#!/bin/bash
# measurement aid: registers / scratch / occupancy of every kernel of libfmarl (compiler remarks, no GPU needed)
cd "$(dirname "$0")/../fair_marl_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -shared -fPIC -Rpass-analysis=kernel-resource-usage -o /tmp/kres_lib.so libfmarl.hip 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); vals={}
    for k in ('VGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','SGPRs Spill'):
        m=re.search(r'    '+k+r': (\d+)',l)
        if m and cur: vals[k]=m.group(1)
    if cur and 'SGPRs Spill' in l and ' Spill: ' in l and 'VGPRs Spill' not in l:
        print('%-90s vgpr %4s scratch %4s occ %s' % (cur[:90], vals.get('VGPRs'), vals.get('ScratchSize \\\\[bytes/lane\\\\]'), vals.get('Occupancy \\\\[waves/SIMD\\\\]')))
"
