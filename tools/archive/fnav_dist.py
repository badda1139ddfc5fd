"""measurement aid (GPU box): per-launch times of fairnav_kernel<true> over 100 steps from a lockstep start (hipEvents per launch):
launches in which no env ends, launches with a reset pass, the launch that ends the episode.  usage: python tools/archive/fnav_dist.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench, fair_marl_amd as fm
spec = bench.CONFIGS['fnav']; cfg = fm.EnvConfig(**spec['env']); n = spec['n_envs']; dev = 'cuda:0'
g = torch.Generator(device=dev); g.manual_seed(1)
tape = torch.randint(0, 5, (25, n, cfg.N), device=dev, generator=g, dtype=torch.int32)
eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, tune_placement=0)
eng.reset()
for t in range(25): eng.step(tape[t])
torch.cuda.synchronize()
eng.profile_enable(100)
for t in range(100): eng.step(tape[t % 25])
torch.cuda.synchronize()
ms = np.array(eng.profile_read())
print('geometry', eng.launch_geometry())
print('mean %.4f median %.4f min %.4f max %.4f' % (ms.mean(), np.median(ms), ms.min(), ms.max()))
print(np.round(ms[:50] * 1000).astype(int).tolist())
