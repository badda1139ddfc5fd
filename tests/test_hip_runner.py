"""GPU parity of the runner-side rows (a8, a14, f-2, f-3, f-4) against fixtures produced by the
REFERENCE itself (tests/golden/gen_runner.py): its vec env with resets on the device's Philox stream,
GMPERunner.warmup / insert into a real GraphReplayBuffer, processAdj, process_infos + metric readers, update_graph.
Everything goes through the C-ABI (RolloutEngine / DeviceRolloutBuffer are thin ctypes callers)."""
import json

import numpy as np
import pytest
import torch

import fair_marl_amd as fm
from oracle import runner_oracle as ro
from helpers import RUNNER, load

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
F32 = dict(rtol=1e-5, atol=1e-5)   # north_star: outputs within 1e-5 in float32


def engine_of(fx, **kw):
    args = json.loads(str(fx['args']))
    cfg = fm.EnvConfig.from_args(args)
    n = fx['obs'].shape[1]
    return fm.RolloutEngine(cfg, n, device=DEV, seed=int(fx['seed']), **kw), cfg, args, n


@pytest.mark.parametrize('geom', [0, 2, 'full'])
@pytest.mark.parametrize('name', RUNNER)
def test_engine_equals_the_reference_on_the_philox_stream(name, geom):
    """No state injection anywhere: device resets (placement, assignment, reset observation) and steps against the
    reference's own outputs through every auto-reset of the run -- at one env per workgroup (the library's choice for so
    few envs), two, and all of the fixture's envs inside one workgroup of the full-batch shape (the staged episode's commit
    and re-emission at env positions > 0, fairnav envs ending at different steps inside one workgroup)."""
    fx = load(name)
    eng, cfg, args, n = engine_of(fx, count_edges=True, envs_per_workgroup=4096 if geom == 'full' else geom)
    if geom == 'full':
        assert eng.envs_per_workgroup >= min(n, 8)
    obs, ids, node, adj = eng.reset()
    # f-3 fused: buffer slot s of episode ep holds the output of step ep * T + s - 1 (slot 0 of episode 0: the reset)
    T = cfg.episode_length
    padj_at = {}
    for key in fx.files:
        if '_padj' in key and key.endswith('_index'):
            ep, slot = int(key[2:key.index('_')]), int(key[key.index('padj') + 4:key.rindex('_')])
            padj_at[ep * T + slot - 1] = key

    def check_fused_edges(t):
        key = padj_at[t]
        ei, ea, off = eng.process_adj(per_agent=True)            # counts from the emission, edges from the state
        assert np.array_equal(ei.cpu().numpy(), fx[key]), key
        np.testing.assert_allclose(ea.cpu().numpy(), fx[key.replace('_index', '_attr')], rtol=1e-6, atol=1e-6)
        assert int(off[-1]) == fx[key].shape[1] and off.numel() == n * cfg.N + 1
    check_fused_edges(-1)
    np.testing.assert_allclose(obs.cpu().numpy(), fx['ep0_obs'][0], **F32)
    np.testing.assert_allclose(node.cpu().numpy(), fx['ep0_node_obs'][0], **F32)
    np.testing.assert_allclose(adj[:, 0].cpu().numpy(), fx['ep0_adj'][0], **F32)
    for t in range(fx['actions'].shape[0]):
        # Scenario.update_graph where MultiAgentGraphEnv.step calls it (environment.py:817-818), float64 from the state
        ei, ew, nnz = (x.cpu().numpy() for x in eng.update_graph())
        for e in range(n):
            k = int(fx['edge_nnz'][t, e])
            assert int(nnz[e]) == k, (t, e)
            assert np.array_equal(ei[e][:, :k], fx['edge_list'][t, e][:, :k]) and (ei[e][:, k:] == -1).all()
            np.testing.assert_allclose(ew[e][:k], fx['edge_weight'][t, e][:k], rtol=1e-9, atol=1e-9)
        onehot = np.eye(5)[fx['actions'][t]]
        res = eng.step(torch.as_tensor(onehot, dtype=torch.float32, device=DEV))
        obs, ids, node, adj, rew, done, info = (x.cpu().numpy() for x in res)
        msg = '%s step %d' % (name, t)
        np.testing.assert_allclose(obs, fx['obs'][t], err_msg=msg, **F32)
        np.testing.assert_allclose(node, fx['node_obs'][t], err_msg=msg, **F32)
        np.testing.assert_allclose(adj[:, 0], fx['adj'][t], err_msg=msg, **F32)
        np.testing.assert_allclose(rew, fx['reward'][t], err_msg=msg, **F32)
        assert np.array_equal(done.astype(bool), fx['done'][t]), msg
        assert np.array_equal(ids, fx['agent_id'][t])
        keys = [str(k) for k in fx['info_keys']]
        slot = dict(fm.infos.key_map(cfg.scenario_name))
        got = np.stack([info[..., slot[k]] for k in keys], axis=-1)
        np.testing.assert_allclose(got, fx['info'][t], err_msg=msg, **F32)
        if t in padj_at:
            check_fused_edges(t)
    assert fx['reset_count'].sum() >= 2 and len(padj_at) >= 4


@pytest.mark.parametrize('name', RUNNER)
def test_span_equals_the_reference_on_the_philox_stream(name):
    """The same reference rollouts through fmarl_step_span: the whole action tape of the fixture in ONE call (runs of steps as
    single launches, the episode ends as launches of their own; the third scenario steps), every step's outputs written
    through per-step strides and compared with the reference's own outputs of that step."""
    fx = load(name)
    eng, cfg, args, n = engine_of(fx)
    T = fx['actions'].shape[0]
    N, E, D, F = cfg.N, cfg.E, cfg.obs_dim, cfg.node_feat
    z = lambda *sh, dt=torch.float32: torch.zeros(*sh, dtype=dt, device=DEV)  # noqa: E731
    big = dict(obs=z(T, n, N, D), node_obs=z(T, n, N, E, F), adj=z(T, n, E, E), reward=z(T, n, N), done=z(T, n, N, dt=torch.uint8),
               info=z(T, 14, n, N))
    eng.reset()
    eng.use_outputs(eng.new_output_set(obs=big['obs'][0], node_obs=big['node_obs'][0], adj_env=big['adj'][0], reward=big['reward'][0],
                                       done=big['done'][0], info_planes=big['info'][0]))
    eng.step_span(torch.as_tensor(fx['actions'], dtype=torch.int32, device=DEV).contiguous(), strides={k: v[0].numel() for k, v in big.items()})
    torch.cuda.synchronize()
    keys = [str(k) for k in fx['info_keys']]
    slot = dict(fm.infos.key_map(cfg.scenario_name))
    for k in ('obs', 'node_obs', 'adj', 'reward'):
        np.testing.assert_allclose(big[k].cpu().numpy(), fx[k], err_msg='%s %s' % (name, k), **F32)
    assert np.array_equal(big['done'].cpu().numpy().astype(bool), fx['done'])
    info = big['info'].permute(0, 2, 3, 1).cpu().numpy()                     # (T, n, N, 14)
    np.testing.assert_allclose(np.stack([info[..., slot[k]] for k in keys], axis=-1), fx['info'], err_msg=name + ' info', **F32)
    assert fx['reset_count'].sum() >= 2


@pytest.mark.parametrize('name', RUNNER)
def test_device_rollout_buffer_equals_the_reference_replay_buffer(name):
    """DeviceRolloutBuffer (filled in place by the kernels) vs GraphReplayBuffer after the reference runner's
    warmup / insert / after_update; processAdj on stored slots; process_infos and the metric readers at episode end."""
    fx = load(name)
    eng, cfg, args, n = engine_of(fx)
    T, N, E = cfg.episode_length, cfg.N, cfg.E
    buf = fm.DeviceRolloutBuffer(eng, episode_length=T)
    buf.reset()
    keys = [str(k) for k in fx['info_keys']]
    for ep in range(fx['actions'].shape[0] // T):
        for s in range(T):
            a = torch.as_tensor(fx['actions'][ep * T + s], dtype=torch.int32, device=DEV)
            buf.insert_step(a)
        for k in ('obs', 'node_obs', 'share_obs', 'rewards'):
            np.testing.assert_allclose(getattr(buf, k).cpu().numpy(), fx['ep%d_%s' % (ep, k)], err_msg='%s ep %d' % (k, ep), **F32)
        np.testing.assert_allclose(buf.adj_env.cpu().numpy(), fx['ep%d_adj' % ep], **F32)
        assert buf.adj.shape == (T + 1, n, N, E, E) and buf.share_obs.shape == fx['ep0_share_obs'].shape
        for k in ('masks', 'active_masks', 'agent_id', 'share_agent_id'):
            got = getattr(buf, k).cpu().numpy()
            assert got.dtype == fx['ep%d_%s' % (ep, k)].dtype and np.array_equal(got, fx['ep%d_%s' % (ep, k)]), (k, ep)
        # f-3: the policy's edges of stored slots (gnn.py:307-326 on np.concatenate(buffer.adj[step]))
        for key in fx.files:
            if key.startswith('ep%d_padj' % ep) and key.endswith('_index'):
                s = int(key[key.index('padj') + 4:key.rindex('_')])
                ei, ea, off = eng.process_adj(adj_env=buf.adj_env[s], per_agent=True)
                assert np.array_equal(ei.cpu().numpy(), fx[key]), key
                np.testing.assert_allclose(ea.cpu().numpy(), fx[key.replace('_index', '_attr')], rtol=1e-6, atol=1e-6)
                assert int(off[-1]) == fx[key].shape[1] and off.numel() == n * N + 1
        # f-4: process_infos on the infos of the episode's last step (graph_mpe_runner.py:146)
        names = [str(x) for x in fx['ep%d_env_info_names' % ep]]
        want = fx['ep%d_env_infos' % ep]                       # (names, N, n); NaN rows = empty lists
        lists = eng.process_infos(reduce=None)
        means = eng.process_infos()
        assert sorted({k.split('/', 1)[1] for k in lists}) == names
        assert list(lists) == list(ro.process_infos(fx['info'][(ep + 1) * T - 1], keys, T))   # dict order of :260-274
        for j, nm in enumerate(names):
            for a in range(N):
                k = 'agent%d/%s' % (a, nm)
                if np.isnan(want[j, a]).all():
                    assert lists[k].numel() == 0 and k not in means
                else:
                    np.testing.assert_allclose(lists[k].cpu().numpy(), want[j, a], err_msg=k, **F32)
                    assert abs(means[k] - want[j, a].mean()) <= 1e-5 * (1 + abs(want[j, a].mean())), k
        for reader in ro.METRIC_PATTERNS:
            key = 'ep%d_%s' % (ep, reader)
            if key in fx.files:
                np.testing.assert_allclose(getattr(eng, reader)(), fx[key], err_msg=reader, **F32)
            else:
                with pytest.raises(IndexError):
                    getattr(eng, reader)()
        buf.after_update()
        for k in ('obs', 'node_obs', 'share_obs'):
            np.testing.assert_allclose(getattr(buf, k)[0].cpu().numpy(), fx['ep%d_after_%s0' % (ep, k)], **F32)
        for k in ('masks', 'active_masks'):
            assert np.array_equal(getattr(buf, k)[0].cpu().numpy(), fx['ep%d_after_%s0' % (ep, k)]), k
    assert (fx['ep0_masks'] == 0).any()


def test_update_graph_and_process_adj_at_the_threshold():
    """edges_kat.npz: distances exactly at max_edge_dist, one float64 ulp beyond / inside it, coincident entities.
    update_graph from the state is bit-exact (float64 weights included); processAdj is strict."""
    fx = load('edges_kat.npz')
    args = json.loads(str(fx['args']))
    cfg = fm.EnvConfig.from_args(args)
    n = fx['ent_pos'].shape[0]
    eng = fm.RolloutEngine(cfg, n, device=DEV, seed=0)
    eng.reset()
    N, L = cfg.N, cfg.num_landmarks
    pos = fx['ent_pos']
    eng.set_state(dict(agent_pos=pos[:, :N], landmark_pos=pos[:, N:N + L], obstacle_pos=pos[:, N + L:]))
    ei, ew, nnz = (x.cpu().numpy() for x in eng.update_graph())
    assert ew.dtype == np.float64
    for c in range(n):
        k = int(fx['edge_nnz'][c])
        assert int(nnz[c]) == k
        assert np.array_equal(ei[c][:, :k], fx['edge_list'][c][:, :k])
        assert np.array_equal(ew[c][:k], fx['edge_weight'][c][:k])     # bit for bit
        pairs = set(map(tuple, ei[c][:, :k].T))
        assert (0, 1) in pairs and (0, 3) in pairs and (0, 2) not in pairs and (0, 4) not in pairs
    d = pos[:, :, None] - pos[:, None]
    adj32 = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32)
    ei, ea, off = eng.process_adj(adj_env=torch.as_tensor(adj32, device=DEV))
    off = off.cpu().numpy()
    for c in range(n):
        k = int(fx['padj_nnz'][c])
        assert off[c + 1] - off[c] == k
        got = ei[:, off[c]:off[c + 1]].cpu().numpy() - c * cfg.E
        assert np.array_equal(got, fx['padj_index'][c][:, :k])
        assert np.array_equal(ea[off[c]:off[c + 1]].cpu().numpy().astype(np.float64), fx['padj_attr'][c][:k])
        assert (0, 1) not in set(map(tuple, got.T))
    # the float32 rule on a float32 matrix is the <= variant of the same kernels
    ei2, ew2, nnz2 = eng.update_graph(adj_env=torch.as_tensor(adj32, device=DEV))
    assert ew2.dtype == torch.float32 and int(nnz2[0]) >= int(fx['padj_nnz'][0])


def _env_fns(args, n, seed, factory):
    """make_train_env's get_env_fn (onpolicy/scripts/train_mpe.py:21-43) with this package's factories."""
    import argparse
    ns = argparse.Namespace(**args)

    def get_env_fn(rank):
        def init_env():
            env = factory(ns)
            env.seed(seed + rank * 1000)
            return env
        return init_env
    return [get_env_fn(r) for r in range(n)]


@pytest.mark.parametrize('name', ['runner_nav.npz', 'runner_fnav.npz', 'runner_form.npz'])
def test_runner_loop_drops_in(name):
    """The reference runner's call sequence (graph_mpe_runner.py:178-203 warmup, :54-100 step loop, :397-436 one-hot
    float64 actions, :438-488 insert, base_runner.py:197-276 process_infos) driven through fair_marl_amd's
    GraphSubprocVecEnv; the buffers it fills and the env_infos it logs equal the reference run's (fixture)."""
    import warnings
    fx = load(name)
    args = json.loads(str(fx['args']))
    n, seed = fx['obs'].shape[1], int(fx['seed'])
    T, N = args['episode_length'], args['num_agents']
    with warnings.catch_warnings():
        warnings.simplefilter('error')          # seed + 1000 r is the convention: no warning
        envs = fm.GraphSubprocVecEnv(_env_fns(args, n, seed, fm.GraphMPEEnv), device=DEV)
    spaces = json.loads(str(fx['spaces']))
    for attr, key in (('observation_space', 'obs'), ('share_observation_space', 'share_obs'), ('node_observation_space', 'node_obs'),
                      ('adj_observation_space', 'adj'), ('agent_id_observation_space', 'agent_id'),
                      ('share_agent_id_observation_space', 'share_agent_id'), ('edge_observation_space', 'edge')):
        sp = getattr(envs, attr)
        assert len(sp) == N and sp[0].__class__.__name__ == 'Box' and list(sp[0].shape) == spaces[key], attr
    assert envs.action_space[0].__class__.__name__ == 'Discrete' and envs.action_space[0].n == spaces['action_n']
    D, E, F = fx['obs'].shape[-1], fx['node_obs'].shape[-2], fx['node_obs'].shape[-1]
    buf = ro.ReplayBuffer(T, n, N, D, E, F)
    obs, agent_id, node_obs, adj = envs.reset()                                  # warmup
    assert obs.dtype == np.float64 and agent_id.dtype == np.int64 and adj.shape == (n, N, E, E) and adj.dtype == np.float64
    # adj: N identical copies per env as a read-only stride-0 view by default (what the runner does with it -- copy(),
    # indexing, concatenate -- is unaffected); materialize_adj gives the reference's N writable copies
    assert not adj.flags.writeable and adj.strides[1] == 0 and adj.copy().flags.writeable
    assert np.concatenate(adj).shape == (n * N, E, E)
    buf.warmup(obs, agent_id, node_obs, adj)
    keys = [str(k) for k in fx['info_keys']]
    for ep in range(fx['actions'].shape[0] // T):
        active_masks = np.ones((n, N, 1), dtype=np.int32)
        for step in range(T):
            actions = fx['actions'][ep * T + step][..., None]                   # (n, N, 1) as the policy returns them
            actions_env = np.squeeze(np.eye(envs.action_space[0].n)[actions], 2)   # :429-430
            res = envs.step(actions_env)
            assert len(res) == 7
            obs, agent_id, node_obs, adj, rewards, dones, infos = res
            assert dones.dtype == np.bool_ and rewards.dtype == np.float64 and isinstance(infos, tuple)
            active_masks[dones] = np.zeros(((dones).astype(int).sum(), 1), dtype=np.float32)       # :66 boolean index
            dones_env = np.all(dones, axis=1)
            active_masks[dones_env] = np.ones(((dones_env).astype(int).sum(), N, 1), dtype=np.float32)
            buf.insert(obs, agent_id, node_obs, adj, rewards, dones)
        for k in ('share_obs', 'obs', 'node_obs', 'rewards'):
            np.testing.assert_allclose(getattr(buf, k), fx['ep%d_%s' % (ep, k)], err_msg='%s ep %d' % (k, ep), **F32)
        np.testing.assert_allclose(buf.adj[:, :, 0], fx['ep%d_adj' % ep], **F32)
        for k in ('agent_id', 'share_agent_id', 'masks', 'active_masks'):
            assert np.array_equal(getattr(buf, k), fx['ep%d_%s' % (ep, k)]), k
        # process_infos exactly as base_runner.py:208-243 walks the structure
        env_infos = {}
        for a in range(N):
            for key, nm in ro.ENV_INFO_NAMES.items():
                vals = []
                for info in infos:
                    if key in info[a].keys():
                        v = info[a][key]
                        if key == 'Time_req_to_goal' and v == -1:
                            v = T * 0.1
                        vals.append(v)
                env_infos['agent%d/%s' % (a, nm)] = vals
        names = [str(x) for x in fx['ep%d_env_info_names' % ep]]
        want = fx['ep%d_env_infos' % ep]
        for j, nm in enumerate(names):
            for a in range(N):
                v = env_infos['agent%d/%s' % (a, nm)]
                if np.isnan(want[j, a]).all():
                    assert v == []
                else:
                    np.testing.assert_allclose(v, want[j, a], err_msg=nm, **F32)
        assert list(infos[0][0].keys())[0] == 'individual_reward' and set(infos[0][0].keys()) == set(keys)
        buf.after_update()
    envs.close()
    full = fm.GraphSubprocVecEnv(_env_fns(args, 2, seed, fm.GraphMPEEnv), device=DEV)
    full.materialize_adj = True
    a4 = full.reset()[3]
    assert a4.flags.writeable and a4.strides[1] != 0 and np.array_equal(a4[:, 0], a4[:, 1])
    full.close()
    # other seeds than seed + 1000 r cannot be honoured per env: say so
    fns = _env_fns(args, 3, seed, fm.GraphMPEEnv)

    def odd():
        env = fm.GraphMPEEnv(__import__('argparse').Namespace(**args))
        env.seed(12345)
        return env
    with pytest.warns(UserWarning, match='only the first env'):
        fm.GraphSubprocVecEnv(fns[:2] + [odd], device=DEV).close()


def test_graph_dummy_vec_env_single_env_across_episode_ends():
    """eval / render path (graph_mpe_runner.py:684-685): GraphDummyVecEnv with ONE env, 8-tuple, reset_count going
    0 -> 1 at the episode end, infos as an object array; data = env 0 of the reference run."""
    fx = load('runner_nav.npz')
    args = json.loads(str(fx['args']))
    seed = int(fx['seed'])
    envs = fm.GraphDummyVecEnv(_env_fns(args, 1, seed, fm.GraphMPEEnv), device=DEV)
    obs, agent_id, node_obs, adj = envs.reset()
    np.testing.assert_allclose(obs, fx['ep0_obs'][0][:1], **F32)
    keys = [str(k) for k in fx['info_keys']]
    counts = []
    for t in range(fx['actions'].shape[0]):
        res = envs.step(np.eye(5)[fx['actions'][t][:1]])
        assert len(res) == 8
        obs, agent_id, node_obs, adj, rewards, dones, infos, reset_count = res
        np.testing.assert_allclose(obs, fx['obs'][t][:1], **F32)
        np.testing.assert_allclose(node_obs, fx['node_obs'][t][:1], **F32)
        np.testing.assert_allclose(adj[:, 0], fx['adj'][t][:1], **F32)
        np.testing.assert_allclose(rewards, fx['reward'][t][:1], **F32)
        assert np.array_equal(dones, fx['done'][t][:1])
        assert isinstance(infos, np.ndarray) and infos.dtype == object and infos.shape == (1, args['num_agents'])
        got = np.array([[infos[0][a][k] for k in keys] for a in range(args['num_agents'])])
        np.testing.assert_allclose(got, fx['info'][t][0], **F32)
        assert reset_count == int(fx['done'][t][0].all())
        counts.append(reset_count)
    assert counts.count(1) == 2 and counts[0] == 0
    envs.close()


@pytest.mark.parametrize('name', ['runner_mpe.npz', 'runner_mpe_collab.npz'])
def test_non_graph_vec_envs_equal_the_reference(name):
    """DummyVecEnv / SubprocVecEnv over MPEEnv (env_name == 'MPE'): reset() -> obs, step() -> (obs, rews, dones, infos),
    against the reference's DummyVecEnv run (incl. auto-resets and the collaborative [[sum]] * N reward shape)."""
    fx = load(name)
    args = json.loads(str(fx['args']))
    n, seed = fx['obs'].shape[1], int(fx['seed'])
    keys = [str(k) for k in fx['info_keys']]
    for cls, as_array in ((fm.DummyVecEnv, True), (fm.SubprocVecEnv, False)):
        envs = cls(_env_fns(args, n, seed, fm.MPEEnv), device=DEV)
        obs = envs.reset()
        assert isinstance(obs, np.ndarray)
        np.testing.assert_allclose(obs, fx['reset_obs'], **F32)
        for t in range(fx['actions'].shape[0]):
            res = envs.step(np.eye(5)[fx['actions'][t]])
            assert len(res) == 4
            obs, rews, dones, infos = res
            np.testing.assert_allclose(obs, fx['obs'][t], **F32)
            assert rews.shape == fx['reward'][t].shape
            np.testing.assert_allclose(rews, fx['reward'][t], **F32)
            assert np.array_equal(dones, fx['done'][t])
            assert (isinstance(infos, np.ndarray) and infos.shape == (n, args['num_agents'])) if as_array else isinstance(infos, tuple)
            got = np.array([[[infos[e][a][k] for k in keys] for a in range(args['num_agents'])] for e in range(n)])
            np.testing.assert_allclose(got, fx['info'][t], **F32)
        envs.close()
    import argparse
    with pytest.raises(NotImplementedError):
        fm.MPEEnv(argparse.Namespace(**dict(args, scenario_name='fair_graph_formation')))


@pytest.mark.parametrize('kw,n', [(dict(num_agents=32, num_landmarks=32, num_obstacles=8), 40),            # 16-byte adj path
                                  (dict(num_agents=5, num_landmarks=5, num_obstacles=2, num_walls=2, max_edge_dist=0.6), 300),
                                  (dict(scenario_name='fair_graph_formation', num_agents=10, num_landmarks=1, num_obstacles=3), 77),
                                  (dict(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=3, num_landmarks=3,
                                        num_obstacles=2, min_dist_thresh=0.3), 200)],
                         ids=['nav32', 'nav5w2', 'formation', 'fairnav'])
def test_fused_process_adj_equals_the_two_pass_result(kw, n):
    """SURVEY section 8 f-3 as specified: counts from the adj emission (step and reset, incl. envs that auto-reset
    inside the step), prefix sum on the device, edges from the world state == processAdj of the emitted float32 matrix
    (oracle.runner_oracle.process_adj, pinned by the reference's processAdj) -- bit for bit, with and without the one
    host read that sizes the result."""
    cfg = fm.EnvConfig(episode_length=7, **kw)
    eng = fm.RolloutEngine(cfg, n, device=DEV, seed=13, count_edges=True)
    g = torch.Generator(device=DEV); g.manual_seed(3)
    eng.reset()
    for t in range(17):   # crosses two episode ends
        if t:
            eng.step(torch.randint(0, 5, (n, cfg.N), device=DEV, generator=g, dtype=torch.int32))
        adj = eng.adj_env.cpu().numpy()
        for per_agent in (False, True):
            batch = np.repeat(adj[:, None], cfg.N, axis=1).reshape(-1, cfg.E, cfg.E) if per_agent else adj
            want_i, want_a = ro.process_adj(batch, cfg.max_edge_dist)
            ei, ea, off = eng.process_adj(per_agent=per_agent)
            assert np.array_equal(ei.cpu().numpy(), want_i) and np.array_equal(ea.cpu().numpy(), want_a), (t, per_agent)
            counts = np.bincount(want_i[0] // cfg.E, minlength=batch.shape[0])
            assert np.array_equal(off.cpu().numpy(), np.concatenate([[0], np.cumsum(counts)]))
        if t % 5 == 0:    # no host read at all: a generous buffer, then a tight one that drops the overflow
            cap = want_i.shape[1] + 100
            ei, ea, off = eng.process_adj(per_agent=True, max_edges=cap)
            k = int(off[-1])
            assert k == want_i.shape[1] and np.array_equal(ei[:, :k].cpu().numpy(), want_i) and np.array_equal(ea[:k].cpu().numpy(), want_a)
            ei, ea, off = eng.process_adj(per_agent=True, max_edges=k // 2)
            assert int(off[-1]) == k and np.array_equal(ei.cpu().numpy(), want_i[:, :k // 2])
    # an engine that does not count falls back to the two passes over adj: same result
    plain, counting = fm.RolloutEngine(cfg, n, device=DEV, seed=13), fm.RolloutEngine(cfg, n, device=DEV, seed=13, count_edges=True)
    plain.reset(); counting.reset()
    a, b, c = plain.process_adj()
    d, e, f = counting.process_adj()
    assert plain.outs.edge_nnz is None and torch.equal(a, d) and torch.equal(b, e) and torch.equal(c, f)


def test_edge_offsets_prefix_sum_at_scale():
    """fmarl_edge_offsets across many chunks (2048 graphs each) and with replication."""
    import ctypes as C
    from fair_marl_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(0)
    for n, reps in ((1, 1), (2047, 1), (2048, 1), (2049, 3), (70001, 1), (5000, 32), (300000, 2)):
        nnz = rs.randint(0, 5000, size=n).astype(np.int32)
        d = torch.as_tensor(nnz, device=DEV)
        off = torch.empty(n * reps + 1, dtype=torch.int64, device=DEV)
        _lib.check(lib.fmarl_edge_offsets(d.data_ptr(), n, reps, off.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'offsets')
        want = np.concatenate([[0], np.cumsum(np.repeat(nnz.astype(np.int64), reps))])
        assert np.array_equal(off.cpu().numpy(), want), (n, reps)


def test_returned_arrays_belong_to_the_caller_unless_reuse_is_opted_into():
    """Default: every array a wrapper returns is fresh -- one the caller still holds (directly, in a list, through a raw pointer)
    is never rewritten by later steps, like the arrays the reference's workers pipe back.  ``reuse_outputs = k`` is the
    explicit opt-in: the arrays of one call come back k calls later (valid until then), values unchanged in between."""
    import argparse
    args = argparse.Namespace(**json.loads(str(load('runner_nav.npz')['args'])))
    fns = _env_fns(vars(args), 6, 3, fm.GraphMPEEnv)
    venv = fm.GraphSubprocVecEnv(fns)
    venv.reset()
    rs = np.random.RandomState(0)
    act = lambda: np.eye(5)[rs.randint(0, 5, size=(6, args.num_agents))]  # noqa: E731
    kept = []
    for t in range(6):
        res = venv.step(act())
        kept.append((res, [np.array(x, copy=True) for x in (res[0], res[2], res[3], res[4])]))
    ids = set()
    for res, copies in kept:
        for x, c in zip((res[0], res[2], res[3], res[4]), copies):
            assert np.array_equal(x, c)            # nothing the caller kept was overwritten
        for x in (res[0], res[2], res[4]):
            assert id(x) not in ids and x.flags.owndata and x.flags.writeable
            ids.add(id(x))
    venv.close()
    venv = fm.GraphSubprocVecEnv(fns)
    venv.reuse_outputs = 2
    venv.reset()
    outs = [venv.step(act()) for _ in range(5)]
    assert outs[0][0] is outs[2][0] is outs[4][0] and outs[1][0] is outs[3][0] and outs[0][0] is not outs[1][0]
    assert outs[0][4] is outs[2][4] and outs[0][4] is not outs[0][5]     # rewards and dones (same shape) have their own rings
    last = np.array(outs[4][2], copy=True)
    venv.step(act())                                                      # writes generation 1: generation 0 stays valid
    assert np.array_equal(outs[4][2], last)
    venv.close()
