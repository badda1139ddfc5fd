"""measurement aid (GPU box): fairnav_kernel<true> launch times once the envs' episodes have drifted apart (a threshold at which
goals are reached, so that episodes end at all phases): usage python tools/archive/fnav_steady.py [min_dist_thresh] [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import fair_marl_amd as fm  # noqa: E402

thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
spec = bench.CONFIGS['fnav']
cfg = fm.EnvConfig(**dict(spec['env'], min_dist_thresh=thr))
n, dev = spec['n_envs'], 'cuda:0'
g = torch.Generator(device=dev); g.manual_seed(1)
tape = torch.randint(0, 5, (50, n, cfg.N), device=dev, generator=g, dtype=torch.int32)
eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, tune_placement=0)
eng.reset()
for t in range(steps):
    eng.step(tape[t % 50])
torch.cuda.synchronize()
ep = eng.field('episode').to(torch.float64)
cs = eng.field('cur_step')
print('after %d steps: episodes per env %.1f, current step spread: %s' % (steps, float(ep.mean()), torch.bincount(cs, minlength=26).tolist()))
eng.profile_enable(200)
for t in range(200):
    eng.step(tape[t % 50])
torch.cuda.synchronize()
ms = np.array(eng.profile_read())
print('launch ms: mean %.4f median %.4f min %.4f max %.4f' % (ms.mean(), np.median(ms), ms.min(), ms.max()))
