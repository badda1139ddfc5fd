"""measurement aid (GPU box): nav_fairassign_fairrew_formation_graph, 65 536 x 3 -- a launch per step against fmarl_step_span
(fairnav_span_kernel: all steps of the tape in one launch), into one output set and into time slots; also checks that both leave
the same state.  usage: python tools/archive/fnav_span_probe.py [steps=100] [min_dist_thresh]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import fair_marl_amd as fm  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
kw, n = dict(bench.CONFIGS['fnav']['env']), bench.CONFIGS['fnav']['n_envs']
if len(sys.argv) > 2:
    kw['min_dist_thresh'] = float(sys.argv[2])
cfg = fm.EnvConfig(**kw)
dev = torch.device('cuda:0')
gen = torch.Generator(device=dev); gen.manual_seed(2)
tape = torch.randint(0, 5, (T, n, cfg.N), device=dev, generator=gen, dtype=torch.int32)


def run(mode, ring):
    eng = fm.RolloutEngine(cfg, n, device=dev, seed=5, tune_placement=0)
    eng.reset()
    r = fm.OutputRing(eng, T) if ring else None
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == 'span':
            if r is not None:
                eng.use_outputs(r.sets[0]); eng.step_span(tape, strides=r.strides)
            else:
                eng.step_span(tape)
        else:
            for t in range(T):
                if r is not None:
                    eng.use_outputs(r.sets[t])
                eng.step(tape[t])
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / T)
    st = eng.get_state()
    return best, st


for ring in (False, True):
    a, sa = run('step', ring)
    b, sb = run('span', ring)
    import numpy as np; same = all(np.array_equal(sa[k], sb[k]) for k in sa)
    print('%s: launch per step %.4f ms per step, span %.4f ms per step; same final state: %s' % ('time slots' if ring else 'one output set', a * 1e3, b * 1e3, same), flush=True)
