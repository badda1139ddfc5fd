"""one-off soak: BASELINE config 3 for many episodes with invariant checks every few hundred steps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fair_marl_amd as fm
cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
n, dev = 65536, 'cuda:0'
eng = fm.RolloutEngine(cfg, n, device=dev, seed=3)
g = torch.Generator(device=dev); g.manual_seed(1)
tape = torch.randint(0, 5, (64, n, 32), device=dev, generator=g, dtype=torch.int32)
eng.reset()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
for t in range(K):
    obs, ids, node, adj, rew, done, info = eng.step(tape[t % 64])
    if (t + 1) % 250 == 0 or t == K - 1:
        a = eng.adj_env
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(info).all(), t
        assert torch.isfinite(node[::97]).all() and torch.isfinite(a).all(), t
        assert torch.equal(a, a.transpose(1, 2)) and bool((a.diagonal(dim1=1, dim2=2) == 0).all()), t
        assert float(rew.min()) >= -2 * cfg.collision_rew - 1e-6 and float(rew.max()) <= cfg.goal_rew + cfg.fair_rew + 1e-6, t
        st = eng.get_state()
        gm = torch.as_tensor(st['goal_match'])
        assert bool((gm.sort(dim=1).values == torch.arange(32)).all()), t
        assert abs(st['agent_pos']).max() < 50 and abs(st['agent_vel']).max() <= cfg.max_speed + 1e-9, t
        print('step %5d ok: episodes %d..%d, mean reward %.4f, max |pos| %.3f' % (t + 1, st['episode'].min(), st['episode'].max(), float(rew.mean()), abs(st['agent_pos']).max()), flush=True)
print('SOAK_OK')
