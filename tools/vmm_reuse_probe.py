"""measurement aid (GPU box): does an array of hipMemCreate pieces (fmarl_ring_alloc) keep what is written into it when an earlier
array was written and freed before it?  It did not while fmarl_ring_free returned the virtual address range (hipMemAddressFree): a
later hipMemAddressReserve hands the same addresses out again and the GPU keeps stale translations for them -- zeroes in up to 70 %
of the new array right after a fill, more arriving seconds later, reads that disagree.  The allocator now keeps freed ranges out
of circulation (libfmarl.hip ring_release).  usage: python tools/vmm_reuse_probe.py"""
import gc
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fair_marl_amd import _lib  # noqa: E402
from fair_marl_amd.engine import alloc_time_slots  # noqa: E402

lib = _lib.load()
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
A0, A, B = (2, 32768, 6, 16, 11), (2, 32768, 10, 16, 12), (2, 32768, 3, 9, 13)


def report(b, tag):
    import time
    flat = b.view(-1)
    t0 = time.perf_counter()
    seen = []
    for wait in (0.0, 0.2, 1.0, 2.0):
        time.sleep(wait)
        seen.append('%.1fs: %d' % (time.perf_counter() - t0, int((flat != 2.0).sum())))
    direct = int((flat.cpu().numpy() != 2.0).sum())              # hipMemcpy straight out of the array of pieces
    via = int((flat.clone().cpu().numpy() != 2.0).sum())          # a kernel copy into a plain allocation first
    again = int((flat != 2.0).sum())
    print('%s: wrong by kernel reads at %s; by a direct copy to the host %d; by clone + copy %d; kernel read afterwards %d (of %d)'
          % (tag, ', '.join(seen), direct, via, again, flat.numel()), flush=True)
    return again + direct


def cycle(sa, sb, tag):
    a, ia = alloc_time_slots(lib, dev, sa, spread=True)
    a.fill_(1.0); torch.cuda.synchronize()
    del a; gc.collect(); torch.cuda.synchronize()
    b, ib = alloc_time_slots(lib, dev, sb, spread=True)
    b.fill_(2.0); torch.cuda.synchronize()
    bad = report(b, tag + ' after the first fill')
    del b; gc.collect(); torch.cuda.synchronize()
    return bad


if __name__ == '__main__':
    bad = cycle(A, B, 'first') + cycle(A, B, 'second') + cycle(A0, A, 'third') + cycle(A, A0, 'fourth')
    print('wrong elements in total: %d' % bad)
