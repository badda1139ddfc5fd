"""measurement aid (GPU box): does the rate of a pure store stream depend on WHEN in a process its buffer was allocated?  A fresh process
allocates a large buffer, streams into it (fmarl_store_stream, scattered 64 KB chunks), frees it (torch.cuda.empty_cache), allocates
and frees a few odd-sized tensors, allocates the large buffer again, ... -- and the same with the buffer kept and re-used.
usage: python tools/archive/alloc_probe.py [GB=60] [rounds=5]"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fair_marl_amd import _lib  # noqa: E402

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib, dev = _lib.load(), torch.device('cuda:0')
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
nbytes = int(gb * 1e9) // (1 << 20) * (1 << 20)


def rate(buf, chunk=1 << 16):
    chunks = nbytes // chunk
    o = int(chunks * 0.6180339887) | 1
    while math.gcd(o, chunks) != 1:
        o += 2
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    call = lambda: _lib.check(lib.fmarl_store_stream(buf.data_ptr(), nbytes, 2, chunk, o, 0, st), 'fmarl_store_stream')  # noqa: E731
    call()
    e0.record()
    for _ in range(3):
        call()
    e1.record(); e1.synchronize()
    return nbytes / (e0.elapsed_time(e1) / 3) / 1e9


keep = torch.empty(nbytes, dtype=torch.uint8, device=dev)
print('kept buffer (first allocation of the process), %.0f GB at %#x: %.3f TB/s' % (gb, keep.data_ptr(), rate(keep)), flush=True)
for r in range(rounds):
    junk = [torch.empty(int((3 + 7 * k) * 1e8) + 4096 * k, dtype=torch.uint8, device=dev) for k in range(6)]
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    print('round %d: fresh buffer at %#x: %.3f TB/s   kept buffer again: %.3f TB/s' % (r, buf.data_ptr(), rate(buf), rate(keep)), flush=True)
    del buf, junk
    torch.cuda.empty_cache()
