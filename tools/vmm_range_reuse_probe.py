"""measurement aid (GPU box): fmarl_ring_alloc into a KEPT address range (fmarl_ring_free keeps the range reserved; the next array of
the same size maps fresh physical pieces into it).  Is that path clean where a range that went back to the runtime was not
(tools/vmm_reuse_probe.py)?  Allocate / fill / check / free the same two sizes over and over and print the allocator's books.
usage: python tools/vmm_range_reuse_probe.py [cycles=12]"""
import ctypes as C
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fair_marl_amd import _lib  # noqa: E402
from fair_marl_amd.engine import alloc_time_slots  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 12
shapes = [(2, 32768, 10, 16, 12), (5, 3, 1048576), (3, 16384, 6, 16, 11)]
stats = (C.c_uint64 * 6)()
for k in range(cycles):
    shape = shapes[k % len(shapes)]
    try:
        t, inter = alloc_time_slots(lib, dev, shape, spread=True)
    except MemoryError as e:
        print('cycle %d shape %s: %s' % (k, shape, e), flush=True)
        continue
    nz = int((t != 0).sum())
    t.fill_(float(k + 1)); torch.cuda.synchronize()
    flat = t.view(-1)
    w0 = int((flat != float(k + 1)).sum())
    time.sleep(0.3)
    w1 = int((flat != float(k + 1)).sum())
    w2 = int((flat.cpu() != float(k + 1)).sum())
    lib.fmarl_ring_stats(stats)
    print('cycle %d shape %s: nonzero at start %d; wrong after fill %d, 0.3 s later %d, on the host %d; stats %s'
          % (k, shape, nz, w0, w1, w2, [int(v) for v in stats]), flush=True)
    del t, flat
    gc.collect(); torch.cuda.synchronize()
