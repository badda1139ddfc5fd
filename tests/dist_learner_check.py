"""Data-parallel learners: every rank holds the rollout buffer of ITS env shard; ``DeviceRolloutBuffer.advantages(group=...)``
adds the ranks' (count, sum, sum of squares) and standardises with the statistics of the whole batch.  Checked against ONE
buffer over all envs on rank 0: the rank's advantages == the unsharded buffer's rows of that shard (to float32 rounding: the
order of the float64 additions differs), compute_returns bit for bit.

Run under ``python -m torch.distributed.run --nproc-per-node W tests/dist_learner_check.py`` (tests/test_hip_learner.py); the
ranks share the GPU and exchange over gloo.  Prints ``DIST_LEARNER_OK world=<W>`` (rank 0)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import fair_marl_amd as fm  # noqa: E402
from fair_marl_amd.rollout_buffer import DeviceRolloutBuffer  # noqa: E402
from fair_marl_amd.sharding import shard_range  # noqa: E402


def filled(cfg, n, device, arrays, lo, hi):
    eng = fm.RolloutEngine(cfg, n, device=device, seed=3, env_offset=lo)
    buf = DeviceRolloutBuffer(eng).attach_policy(hidden_size=4)
    for k, v in arrays.items():
        getattr(buf, k).copy_(torch.as_tensor(v[:, lo:hi], device=device))
    return buf


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    torch.cuda.set_device(device)
    dist.init_process_group('gloo')
    cfg = fm.EnvConfig(num_agents=4, num_landmarks=4, num_obstacles=2, episode_length=9)
    per, T, N = 57, cfg.episode_length, cfg.N
    n_total = per * world
    rs = np.random.RandomState(12)     # the same arrays on every rank
    f32 = np.float32
    arrays = dict(rewards=(3 * rs.randn(T, n_total, N, 1)).astype(f32), value_preds=rs.randn(T + 1, n_total, N, 1).astype(f32),
                  masks=(rs.rand(T + 1, n_total, N, 1) > 0.1).astype(f32), active_masks=(rs.rand(T + 1, n_total, N, 1) > 0.25).astype(f32))
    nv = rs.randn(n_total, N, 1).astype(f32)
    norm = (0.4, 1.7)
    lo, hi = shard_range(n_total, world, rank)
    buf = filled(cfg, per, device, arrays, lo, hi)
    buf.compute_returns(torch.as_tensor(nv[lo:hi], device=device), norm)
    adv = buf.advantages(norm, group=True)
    stats = buf.advantage_stats.cpu().numpy().copy()
    gathered = [None] * world
    dist.gather_object((adv.cpu().numpy(), buf.returns.cpu().numpy(), stats), gathered if rank == 0 else None, dst=0)
    if rank == 0:
        full = filled(cfg, n_total, device, arrays, 0, n_total)
        full.compute_returns(torch.as_tensor(nv, device=device), norm)
        want = full.advantages(norm).cpu().numpy()
        want_stats = full.advantage_stats.cpu().numpy()
        ret = full.returns.cpu().numpy()
        for r, (a, rr, st) in enumerate(gathered):
            l, h = shard_range(n_total, world, r)
            assert np.array_equal(rr, ret[:, l:h]), 'returns of rank %d' % r
            np.testing.assert_allclose(st, want_stats, rtol=1e-6, err_msg='statistics of rank %d' % r)
            np.testing.assert_allclose(a, want[:, l:h], rtol=2e-6, atol=2e-6, err_msg='advantages of rank %d' % r)
        local = buf.advantages(norm).cpu().numpy()     # without the group: the shard's own statistics -- different numbers
        assert np.abs(local - gathered[0][0]).max() > 1e-4
        print('DIST_LEARNER_OK world=%d' % world, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
