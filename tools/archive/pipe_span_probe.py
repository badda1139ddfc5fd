"""measurement aid (GPU box): k sub-batches on k streams, each running its steps as spans.  usage: python tools/archive/pipe_span_probe.py <config> <k>"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import fair_marl_amd as fm  # noqa: E402

name, k = sys.argv[1], int(sys.argv[2])
spec = bench.CONFIGS[name]
cfg = fm.EnvConfig(**spec['env'])
n, dev, T = spec['n_envs'], 'cuda:0', cfg.episode_length
hint = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pipe = fm.PipelinedRollout(cfg, n, k=k, device=dev, seed=1, tune_placement=0, envs_per_workgroup=hint)
g = torch.Generator(device=dev); g.manual_seed(1)
tapes = [torch.randint(0, 5, (T, n // k, cfg.N), device=dev, generator=g, dtype=torch.int32) for _ in range(k)]
pipe.reset()
pipe.synchronize()


def episode(span):
    for j, (e, s) in enumerate(zip(pipe.engines, pipe.streams)):
        with torch.cuda.stream(s):
            if span:
                e.step_span(tapes[j])
            else:
                for t in range(T):
                    e.step(tapes[j][t])


for span in (False, True, False, True):
    episode(span); pipe.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for ep in range(8):
        episode(span)
    pipe.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%s k=%d %s: %.4f ms per step  %.3e agent-steps/s' % (name, k, 'spans' if span else 'steps', dt / (8 * T) * 1e3, n * cfg.N * 8 * T / dt), flush=True)
