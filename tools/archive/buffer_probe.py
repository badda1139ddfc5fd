"""measurement aid (GPU box): DeviceRolloutBuffer at full size -- allocation time, ms per step of an episode inserted as one span + the
episode-ending step, and as 25 insert_step calls (a policy in the loop: step kernel + the masks of the runner's insert).
usage: python tools/archive/buffer_probe.py [config=cfg3]      (FMARL_RING_SPREAD=0: plain allocations)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import fair_marl_amd as fm  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
dev = 'cuda:0'
cfg = fm.EnvConfig(**bench.CONFIGS[name]['env'])
n = bench.CONFIGS[name]['n_envs']
eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, tune_placement=0)
t0 = time.perf_counter()
buf = fm.DeviceRolloutBuffer(eng)
torch.cuda.synchronize()
print('%s: buffer of %d slots allocated + zeroed in %.2f s; free %.1f GB' % (name, buf.T + 1, time.perf_counter() - t0, torch.cuda.mem_get_info()[0] / 1e9), flush=True)
g = torch.Generator(device=dev); g.manual_seed(0)
ep = cfg.episode_length
tape = torch.randint(0, 5, (ep, n, cfg.N), device=dev, generator=g, dtype=torch.int32)
buf.reset()
for rnd in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    buf.insert_span(tape[:ep - 1]); buf.insert_step(tape[ep - 1])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('episode as insert_span + insert_step: %.4f ms per step' % (dt / ep * 1e3), flush=True)
    buf.after_update()
buf.reset()
for rnd in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(ep):
        buf.insert_step(tape[t])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('episode as %d insert_step calls (a policy in the loop): %.4f ms per step' % (ep, dt / ep * 1e3), flush=True)
    buf.after_update()
