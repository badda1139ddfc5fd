"""measurement aid (GPU box): bench.py's four `secondary` entries of nav_fairassign_fairrew_formation_graph (one launch per step / span,
lockstep start / episodes ending at all phases), twice; FMARL_LIB selects a library variant.  usage: python tools/fnav_lines.py"""
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
for rep in range(2):
    for mode in ('eager', 'span', 'steady', 'steady-span'):
        d = bench.secondary_line('fnav', mode, dev)
        print(mode, 'ms_per_step %.4f kernel %.4f frac %.3f launches %d kernel=%s' % (d['ms_per_step'], d['kernel_avg_ms'], d['frac'], d['kernel_launches'], d['kernel']), flush=True)
