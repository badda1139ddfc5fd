"""measurement aid (GPU box): envs per workgroup of the span launches when every step writes its own time slot (bench.py's default):
the 300-step secondary line of a config at several geometries.  usage: python tools/ring_epb.py <config> <epb,epb,...> [mode=span]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name, epbs = sys.argv[1], [int(v) for v in sys.argv[2].split(',')]
mode = sys.argv[3] if len(sys.argv) > 3 else 'span'
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
for epb in epbs:
    (bench.SPAN_EPB if mode.startswith('span') else bench.EAGER_EPB)[name] = epb
    d = bench.secondary_line(name, mode, dev, steps=150, warmup=25)
    print('%s %s epb hint %d -> %d envs per workgroup: kernel %.4f ms per step, frac %.3f, of the store ceiling %.3f (%.4f ms)'
          % (name, mode, epb, d['envs_per_workgroup'], d['kernel_avg_ms'], d['frac'], d['frac_of_box_ceiling'] or 0, d['store_ceiling_ms'] or 0), flush=True)
