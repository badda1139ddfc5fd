"""measurement aid (GPU box): the box's write ceiling -- pure 16-byte store streams (fmarl_store_stream) over one step's byte
count and over a span's (T steps in distinct slots), in every shape / chunk size / chunk order / workgroup lifetime.

usage: python tools/store_ceiling.py [config=cfg3] [T=24]"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import fair_marl_amd as fm  # noqa: E402
from fair_marl_amd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
spec = bench.CONFIGS[name]
cfg = fm.EnvConfig(**spec['env'])
n, dev = spec['n_envs'], torch.device('cuda:0')
step_bytes = int(bench.algorithmic_bytes(cfg) * n * cfg.N) // 16 * 16
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def scatter(chunks):
    o = int(chunks * 0.6180339887) | 1
    while math.gcd(o, chunks) != 1:
        o += 2
    return o


def time_one(buf, nbytes, shape, chunk, order, persist, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    call = lambda: _lib.check(lib.fmarl_store_stream(buf.data_ptr(), nbytes, shape, chunk, order, persist, st), 'fmarl_store_stream')  # noqa: E731
    call()
    e0.record()
    for _ in range(reps):
        call()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


for label, nbytes, reps in (('one step', step_bytes, 8), ('%d steps' % T, step_bytes * T, 2)):
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    best = (1e9, None)
    ms = time_one(buf, nbytes, 0, 0, 1, 0, reps)
    print('%-9s shape 0 flat                                   %9.4f ms  %.3f TB/s' % (label, ms, nbytes / ms / 1e9), flush=True)
    best = min(best, (ms, 'flat'))
    for shape in (1, 2):
        for chunk in (1 << 16, 1 << 18, 1 << 20, 8 * 122 * 1024, 1 << 22):
            chunks = (nbytes // 16 + chunk // 16 - 1) // (chunk // 16)
            for order in (1, scatter(chunks)):
                for persist in (0, 768, 1024, 2048):
                    if persist >= chunks:
                        continue
                    ms = time_one(buf, nbytes, shape, chunk, order, persist, reps)
                    tag = 'shape %d chunk %8d order %-8s persist %4d' % (shape, chunk, 'dispatch' if order == 1 else 'scatter', persist)
                    print('%-9s %s  %9.4f ms  %.3f TB/s' % (label, tag, ms, nbytes / ms / 1e9), flush=True)
                    best = min(best, (ms, tag))
    print('%-9s BEST %s: %.4f ms = %.4f ms per step = %.3f TB/s = %.3f of 8 TB/s' % (label, best[1], best[0], best[0] * step_bytes / nbytes,
                                                                                nbytes / best[0] / 1e9, nbytes / best[0] / 1e9 / 8000), flush=True)
    del buf
    torch.cuda.empty_cache()
