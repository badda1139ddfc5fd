"""One PPO iteration's data path on the device, end to end, with a stand-in policy:

    rollout (step kernels write the buffer's slots in place)  ->  compute_returns  ->  advantages  ->  minibatches

i.e. what the reference's runner does between ``collect`` and ``ppo_update`` (onpolicy/runner/shared/graph_mpe_runner.py:69-176
run(): warmup, collect / envs.step / insert for episode_length steps, compute(), train()) with NumPy on the host.  The policy
and the PPO update are out of scope for this package: here a random "policy" supplies values / actions / log-probabilities.

    python examples/rollout_to_minibatches.py [n_envs] [num_agents]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fair_marl_amd as fm  # noqa: E402
from fair_marl_amd.rollout_buffer import DeviceRolloutBuffer  # noqa: E402


def main(n_envs=4096, num_agents=3, iterations=2, num_mini_batch=2, data_chunk_length=10, hidden_size=64, device='cuda:0', verbose=True):
    cfg = fm.EnvConfig(num_agents=num_agents, num_landmarks=num_agents, num_obstacles=3)
    eng = fm.RolloutEngine(cfg, n_envs, device=device, seed=0)
    buf = DeviceRolloutBuffer(eng).attach_policy(act_dim=1, recurrent_N=1, hidden_size=hidden_size)
    T, N = buf.T, cfg.N
    gen = torch.Generator(device=device); gen.manual_seed(0)
    rnd = lambda *shape: torch.randn(*shape, device=device, generator=gen)  # noqa: E731
    buf.reset()                                                   # GMPERunner.warmup
    seen = 0
    for it in range(iterations):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(T):                                        # collect + envs.step + insert
            actions = torch.randint(0, 5, (n_envs, N), device=device, generator=gen, dtype=torch.int32)   # the "policy"
            buf.insert_policy(t, rnd(n_envs, N, 1), actions.to(torch.float32), rnd(n_envs, N, 1), rnd(n_envs, N, 1, hidden_size),
                              rnd(n_envs, N, 1, hidden_size))
            buf.insert_step(actions)
        buf.compute_returns(rnd(n_envs, N, 1), (0.0, 1.0), gamma=0.99, gae_lambda=0.95)   # GMPERunner.compute (ValueNorm at its start)
        adv = buf.advantages((0.0, 1.0))                          # GR_MAPPO.train, graph_mappo.py:294-304
        rows = 0
        for sample in buf.recurrent_generator(adv, num_mini_batch, data_chunk_length):   # ... its minibatch loop
            share_obs, obs, node_obs, adj = sample[:4]
            rows += obs.shape[0]                                  # (a real trainer calls ppo_update(sample) here)
        buf.after_update()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        seen += rows
        if verbose:
            print('iteration %d: %d env steps, %d minibatch rows (obs %s, node_obs %s, adj %s) in %.1f ms; mean reward %.3f'
                  % (it, T * n_envs, rows, tuple(obs.shape[1:]), tuple(node_obs.shape[1:]), tuple(adj.shape[1:]), 1e3 * dt,
                     float(buf.rewards.mean())))
    return seen, buf


if __name__ == '__main__':
    a = [int(x) for x in sys.argv[1:3]]
    main(*a)
