"""measurement aid (GPU box): the learner-side kernels of the rollout buffer (fmarl_compute_returns, fmarl_advantages,
fmarl_minibatch_gather) timed with HIP events on the current stream against their algorithmic bytes.
usage: python tools/archive/learner_probe.py [n_envs] [num_agents]   (navigation_graph shapes, T = 25)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fair_marl_amd as fm  # noqa: E402
from fair_marl_amd.rollout_buffer import DeviceRolloutBuffer  # noqa: E402


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    O = {3: 3, 10: 3, 32: 8}.get(N, 3)
    T = 25 if N <= 10 else 4          # 32 agents: a 25-step buffer of 65 536 envs is 216 GB; four steps show the rates
    cfg = fm.EnvConfig(num_agents=N, num_landmarks=N, num_obstacles=O, episode_length=T)
    eng = fm.RolloutEngine(cfg, n, device='cuda:0', seed=3, tune_placement=0)
    buf = DeviceRolloutBuffer(eng).attach_policy(hidden_size=64)
    gen = torch.Generator(device='cuda:0'); gen.manual_seed(1)
    buf.reset()
    buf.insert_span(torch.randint(0, 5, (T, n, N), device='cuda:0', generator=gen, dtype=torch.int32))
    buf.value_preds.normal_(generator=gen)
    nv = torch.randn(n, N, 1, device='cuda:0', generator=gen)
    cells = T * n * N
    E, F, D = cfg.E, cfg.node_feat, cfg.obs_dim
    for gae, proper, norm in ((1, 0, (0.3, 1.7)), (1, 1, (0.3, 1.7)), (0, 0, None)):
        ms = timed(lambda: buf.compute_returns(nv, norm, 0.99, 0.95, bool(gae), bool(proper)))
        arrays = (4 if gae else 3) + (1 if proper else 0) + (1 if (proper and not gae) else 0)   # r, v, m (+ bad) read, returns written
        by = cells * 4 * arrays
        print('compute_returns gae=%d proper=%d norm=%d: %.4f ms  %.0f GB/s (%d arrays x %.1f MB)' % (gae, proper, norm is not None, ms, by / ms / 1e6, arrays, cells * 4 / 1e6))
    ms = timed(lambda: buf.advantages((0.3, 1.7)))
    print('advantages: %.4f ms  %.0f GB/s (returns, value_preds, active_masks read; adv written, re-read, re-written)' % (ms, cells * 4 * 6 / ms / 1e6))
    adv = buf.advantages((0.3, 1.7))
    rows = min(cells, 1 << 20 if N <= 10 else 1 << 17)
    perm = torch.randperm(cells, device='cuda:0', generator=gen)[:rows].contiguous()
    row_bytes = 4 * (N * D + D + E * F + E * E + 1 + N + 2 * 64 + 1 + 1 + 1 + 1 + 1 + 1 + 1 + 5)
    for name, fields in (('all 16 arrays', None), ('without adj / share_obs (env_slot instead)', tuple(k for k in buf.GENERATOR_FIELDS if k not in ('adj', 'share_obs')))):
        rb = row_bytes if fields is None else row_bytes - 4 * (E * E + N * D)
        ms = timed(lambda: list(buf._generate(adv, perm, rows, 1, 0, 1, fields, fields is not None)), reps=5)
        print('feed-forward minibatch of %d rows, %s: %.4f ms  %.0f GB/s read + written (%d B per row)' % (rows, name, ms, 2 * rows * rb / ms / 1e6, rb))
    L = 10 if T >= 10 else 2
    chunks = rows // L
    cperm = torch.randperm(cells // L, device='cuda:0', generator=gen)[:chunks].contiguous()
    ms = timed(lambda: list(buf._generate(adv, cperm, chunks, 1, 1, L, None, False)), reps=5)
    print('recurrent minibatch of %d chunks x %d: %.4f ms  %.0f GB/s read + written' % (chunks, L, ms, 2 * chunks * L * row_bytes / ms / 1e6))


if __name__ == '__main__':
    main()
