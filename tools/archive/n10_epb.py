"""measurement aid (GPU box): n10 (10 agents x 65 536 envs) one launch per step and as spans against envs per workgroup, fresh engines
with the placement probe on, two rounds.  usage: python tools/archive/n10_epb.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench, fair_marl_amd as fm
spec = bench.CONFIGS['n10']; cfg = fm.EnvConfig(**spec['env']); n = spec['n_envs']; dev = 'cuda:0'; T = cfg.episode_length
g = torch.Generator(device=dev); g.manual_seed(1)
tape = torch.randint(0, 5, (T, n, cfg.N), device=dev, generator=g, dtype=torch.int32)
for rnd in range(2):
    for hint in (0, 22, 20, 16):
        eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, envs_per_workgroup=hint)   # placement probe on (default)
        eng.reset()
        for t in range(T): eng.step(tape[t])
        res = []
        for label, fn in (('per step', lambda: [eng.step(tape[t]) for t in range(T)]), ('span', lambda: eng.step_span(tape))):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for ep in range(8): fn()
            torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / (8 * T) * 1e3)
        print('n10 epb=%2d (%d B LDS): per step %.4f ms  span %.4f ms' % (eng.envs_per_workgroup, eng.launch_geometry()[2], res[0], res[1]), flush=True)
        eng.close(); del eng; torch.cuda.empty_cache()
