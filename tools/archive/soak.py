"""one-off soak: a BASELINE-size batch for many episodes with invariant checks every few hundred steps.
    python tools/soak.py [cfg3|cfg4|fnav] [steps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fair_marl_amd as fm
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
cfg = {'cfg3': fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8),
       'cfg4': fm.EnvConfig(scenario_name='fair_graph_formation', num_agents=10, num_landmarks=1, num_obstacles=3),
       'fnav': fm.EnvConfig(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=3, num_landmarks=3, num_obstacles=3,
                            min_dist_thresh=0.2)}[name]
n, dev, N = 65536, 'cuda:0', cfg.N
eng = fm.RolloutEngine(cfg, n, device=dev, seed=3, count_edges=True, emit_graph_record=True)
g = torch.Generator(device=dev); g.manual_seed(1)
tape = torch.randint(0, 5, (64, n, N), device=dev, generator=g, dtype=torch.int32)
eng.reset()
rec = eng.pack_episode()
for t in range(K):
    obs, ids, node, adj, rew, done, info = eng.step(tape[t % 64])
    if (t + 1) % 250 == 0 or t == K - 1:
        a = eng.adj_env
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(info).all(), t
        assert torch.isfinite(node[::97]).all() and torch.isfinite(a).all(), t
        assert torch.equal(a, a.transpose(1, 2)) and bool((a.diagonal(dim1=1, dim2=2) == 0).all()), t
        assert float(rew.min()) >= -2 * cfg.collision_rew - 1e-6 and float(rew.max()) <= cfg.goal_rew + cfg.fair_rew + 1e-6, t
        st = eng.get_state()
        if name != 'cfg4':
            gm = torch.as_tensor(st['goal_match'])
            assert bool((gm.sort(dim=1).values == torch.arange(N)).all()), t
        else:
            assert set(map(float, set(st['slot_occ'].ravel()))) <= {0.0, 1.0}, t
        if name == 'fnav':
            assert set(map(float, set(st['status'].ravel()))) <= {0.0, 1.0}, t
        assert abs(st['agent_pos']).max() < 50 and abs(st['agent_vel']).max() <= cfg.max_speed + 1e-9, t
        assert int(st['place_fails'].sum()) == 0, t
        # the fused policy-edge count agrees with a count over the emitted matrix
        cnt = ((a > 0) & (a < cfg.max_edge_dist)).sum(dim=(1, 2)).to(torch.int32)
        assert torch.equal(cnt, eng.outs.edge_nnz), t
        # the learner-side rebuild from the records equals what the step kernel emitted
        eng.pack_episode(out=rec)
        node2, adj2 = eng.rebuild_graph(obs if eng.step_record_words == 0 else None, rec, step_record=eng.graph_record)
        assert torch.equal(node2, node) and torch.equal(adj2, a), t
        print('step %5d ok: episodes %d..%d, mean reward %.4f, max |pos| %.3f' % (t + 1, st['episode'].min(), st['episode'].max(), float(rew.mean()), abs(st['agent_pos']).max()), flush=True)
print('SOAK_OK', name)
