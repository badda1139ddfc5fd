"""measurement aid (GPU box): step-kernel time against envs per workgroup (FmarlConfig.envs_per_workgroup).
usage: python tools/archive/epb_sweep.py <config> <epb> [<epb> ...]      (0 = the library's choice)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import fair_marl_amd as fm  # noqa: E402

name = sys.argv[1]
spec = bench.CONFIGS[name]
cfg = fm.EnvConfig(**spec['env'])
n, dev = spec['n_envs'], 'cuda:0'
g = torch.Generator(device=dev); g.manual_seed(1)
tape = torch.randint(0, 5, (25, n, cfg.N), device=dev, generator=g, dtype=torch.int32)
for rnd in range(2):
    for hint in [int(a) for a in sys.argv[2:]]:
        eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, tune_placement=0, envs_per_workgroup=hint)
        eng.reset()
        for t in range(25):
            eng.step(tape[t])
        torch.cuda.synchronize()
        eng.profile_enable(200)
        for t in range(200):
            eng.step(tape[t % 25])
        torch.cuda.synchronize()
        ms = eng.profile_read()
        print('%s epb hint %3d -> %3d envs per workgroup: kernel %.4f ms (median %.4f)' % (name, hint, eng.envs_per_workgroup, np.mean(ms), np.median(ms)), flush=True)
        eng.close(); del eng
        torch.cuda.empty_cache()
