"""measurement aid (GPU box): fmarl_step_span (one launch per run of steps between episode ends) against one launch per step.
usage: python tools/archive/span_probe.py <config> [episodes]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import fair_marl_amd as fm  # noqa: E402

name = sys.argv[1]
episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hint = int(sys.argv[3]) if len(sys.argv) > 3 else 0
spec = bench.CONFIGS[name]
cfg = fm.EnvConfig(**spec['env'])
n, dev, T = spec['n_envs'], 'cuda:0', cfg.episode_length
g = torch.Generator(device=dev); g.manual_seed(1)
tape = torch.randint(0, 5, (T, n, cfg.N), device=dev, generator=g, dtype=torch.int32)
a = fm.RolloutEngine(cfg, n, device=dev, seed=1, tune_placement=0, envs_per_workgroup=hint)
b = fm.RolloutEngine(cfg, n, device=dev, seed=1, tune_placement=0, envs_per_workgroup=hint)
name = '%s epb=%d' % (name, b.envs_per_workgroup)
a.reset(); b.reset()
for ep in range(2):
    for t in range(T):
        a.step(tape[t])
    b.step_span(tape)
torch.cuda.synchronize()
sa, sb = a.get_state(), b.get_state()
same = all(np.array_equal(sa[k], sb[k]) for k in sa) and all(torch.equal(getattr(a, k), getattr(b, k)) for k in ('obs', 'node_obs', 'adj_env', 'reward', 'done', 'info'))
print('%s: span == per-step launches (state + outputs, bit for bit): %s' % (name, same))
for rnd in range(2):
    for label, eng, fn in (('per step', a, lambda e: [e.step(tape[t]) for t in range(T)]), ('span', b, lambda e: e.step_span(tape))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ep in range(episodes):
            fn(eng)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('%s %-9s %.4f ms per step  %.3e agent-steps/s' % (name, label, dt / (episodes * T) * 1e3, n * cfg.N * episodes * T / dt), flush=True)
