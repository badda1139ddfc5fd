"""measurement aid (GPU box): does the span kernel's store stream reach HBM, or is part of it absorbed on the die?

A span launch with stride 0 rewrites each workgroup's outputs once per step (the same 122 KB per env, 19-24 times per launch); with
per-step strides every step has a time slot of its own (T, n, ...) and every byte is written once per launch.  Same box, same
process: pure store streams over one step's byte count (fmarl_store_stream), one launch per step, spans at several envs per
workgroup -- each with the same slot every step and with distinct slots.

usage: python tools/span_slots.py [config=cfg3] [episodes=6] [epb list, e.g. 3,4,5,6,8]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import fair_marl_amd as fm  # noqa: E402
from fair_marl_amd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
epbs = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [3, 4, 5, 6, 8]
spec = bench.CONFIGS[name]
cfg = fm.EnvConfig(**spec['env'])
n, dev, ep = spec['n_envs'], torch.device('cuda:0'), cfg.episode_length
N, E, D, F = cfg.N, cfg.E, cfg.obs_dim, cfg.node_feat
T = ep - 1
B = bench.algorithmic_bytes(cfg) * n * N          # algorithmic bytes of one step
lib = _lib.load()
g = torch.Generator(device=dev); g.manual_seed(1)
tape = torch.randint(0, 5, (ep, n, N), device=dev, generator=g, dtype=torch.int32)


def ms_of(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


# ---- 1. pure store streams over one step's byte count
nbytes = int(B) // 16 * 16
buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
best = None
import math
for shape, chunk, scat in ((0, 0, 0), (1, 1 << 20, 0), (1, 1 << 20, 1), (2, 8 * 122 * 1024, 0), (2, 8 * 122 * 1024, 1), (2, 1 << 16, 1)):
    order = 1
    if scat:
        chunks = (nbytes // 16 + chunk // 16 - 1) // (chunk // 16)
        order = int(chunks * 0.6180339887) | 1
        while math.gcd(order, chunks) != 1:
            order += 2
    ms = ms_of(lambda: _lib.check(lib.fmarl_store_stream(buf.data_ptr(), nbytes, shape, chunk, order, 0, st), 'store_stream'), 8)
    print('store stream shape %d chunk %8d B order %d: %.4f ms  %.3f TB/s' % (shape, chunk, order, ms, nbytes / ms / 1e9), flush=True)
    best = ms if best is None else min(best, ms)
print('best pure store stream over %.3f GB: %.4f ms = %.3f of 8 TB/s' % (nbytes / 1e9, best, nbytes / best / 1e6 / 8000), flush=True)
del buf
torch.cuda.empty_cache()

# ---- 2. the slots: (T, n, ...) arrays, every step of a span its own
node_T = torch.empty(T, n, N, E, F, dtype=torch.float32, device=dev)
adj_T = torch.empty(T, n, E, E, dtype=torch.float32, device=dev)
obs_T = torch.empty(T, n, N, D, dtype=torch.float32, device=dev)
rew_T = torch.empty(T, n, N, dtype=torch.float32, device=dev)
done_T = torch.empty(T, n, N, dtype=torch.uint8, device=dev)
info_T = torch.empty(T, 14, n, N, dtype=torch.float32, device=dev)
strides = dict(obs=obs_T[0].numel(), node_obs=node_T[0].numel(), adj=adj_T[0].numel(), reward=n * N, done=n * N, info=info_T[0].numel())
print('slots: %.1f GB' % (sum(t.numel() * t.element_size() for t in (node_T, adj_T, obs_T, rew_T, done_T, info_T)) / 1e9), flush=True)


def run(eng, mode, slots):
    sets = None
    if slots:
        sets = [eng.new_output_set(obs=obs_T[t], reward=rew_T[t], done=done_T[t], node_obs=node_T[t], adj_env=adj_T[t], info_planes=info_T[t])
                for t in range(T)]
    eng.reset()

    def episode():
        if mode == 'span':
            if slots:
                eng.use_outputs(sets[0])
            eng.step_span(tape[:T], strides=strides if slots else None)
            eng.step(tape[T])
        else:
            for t in range(ep):
                if slots:
                    eng.use_outputs(sets[t % T])
                eng.step(tape[t])
    episode()
    torch.cuda.synchronize()
    eng.profile_enable(episodes * ep)
    for _ in range(episodes):
        episode()
    torch.cuda.synchronize()
    ms, steps = eng.profile_read(with_steps=True)
    ms, steps = np.asarray(ms), np.asarray(steps)
    sel = steps > 1 if mode == 'span' else steps == 1
    per_step = ms[sel].sum() / steps[sel].sum()
    return per_step, ms[sel] / steps[sel]


for mode, epb in [('eager', 0)] + [('span', e) for e in epbs]:
    eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, tune_placement=0, envs_per_workgroup=epb)
    for slots in (False, True):
        per_step, each = run(eng, mode, slots)
        print('%-5s epb %d  %-14s %.4f ms per step  frac %.3f  (of best store stream %.3f)  launches min %.4f max %.4f'
              % (mode, eng.envs_per_workgroup, 'distinct slots' if slots else 'same slot', per_step, B / per_step / 1e6 / 8000, best / per_step,
                 each.min(), each.max()), flush=True)
    eng.close()
    del eng
    torch.cuda.empty_cache()
