"""Pins the oracles against the runner-side fixtures (tests/golden/gen_runner.py: the REFERENCE's vec env,
GMPERunner.warmup / insert, GraphReplayBuffer, process_infos + metric readers, update_graph and processAdj, with the
reference's resets drawing from the device's Philox stream).  CPU only."""
import json

import numpy as np
import pytest

from oracle import fairnav_oracle as fnv, formation_oracle as fo, nav_oracle as no, runner_oracle as ro
from oracle.philox import PhiloxStream
from helpers import load, runner_oracle_env, RUNNER

TOL = dict(rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize('name', RUNNER)
def test_oracle_on_the_philox_stream_reproduces_the_reference_rollout(name):
    """Reference resets (random_scenario on the Philox stream) == oracle resets on the same stream, through every
    auto-reset of the run: the a14 pin that does not go through MT19937."""
    fx = load(name)
    env, cfg, keys = runner_oracle_env(fx)
    n = fx['obs'].shape[1]
    obs, ids, node, adj = env.reset()
    np.testing.assert_allclose(obs.astype(np.float32), fx['ep0_obs'][0], rtol=1e-6, atol=1e-6)   # warmup slot (float32 buffer)
    for t in range(fx['actions'].shape[0]):
        o, i, nd, ad, r, d, info, rc = env.step(fx['actions'][t])
        for got, key in ((o, 'obs'), (nd, 'node_obs'), (ad[:, 0], 'adj'), (r, 'reward'), (info, 'info')):
            np.testing.assert_allclose(got, fx[key][t], err_msg='%s step %d %s' % (name, t, key), **TOL)
        assert np.array_equal(i, fx['agent_id'][t])
        assert np.array_equal(d, fx['done'][t]) and rc == fx['reset_count'][t], t
    assert fx['reset_count'].sum() >= 2


@pytest.mark.parametrize('name', RUNNER)
def test_replay_buffer_oracle_equals_graph_replay_buffer(name):
    """runner_oracle.ReplayBuffer fed with the fixture's env outputs == the reference's GraphReplayBuffer after
    GMPERunner.insert, every episode, and after after_update."""
    fx = load(name)
    args = json.loads(str(fx['args']))
    T, N = args['episode_length'], args['num_agents']
    n = fx['obs'].shape[1]
    D, E, F = fx['obs'].shape[-1], fx['node_obs'].shape[-2], fx['node_obs'].shape[-1]
    buf = ro.ReplayBuffer(T, n, N, D, E, F)
    adj_of = lambda a: np.broadcast_to(a[:, None], (n, N, E, E))  # noqa: E731
    # slot 0 of the first episode is the reset observation the reference's warmup stored
    buf.warmup(fx['ep0_obs'][0], fx['ep0_agent_id'][0], fx['ep0_node_obs'][0], adj_of(fx['ep0_adj'][0]))
    episodes = fx['actions'].shape[0] // T
    for ep in range(episodes):
        for s in range(T):
            t = ep * T + s
            buf.insert(fx['obs'][t], fx['agent_id'][t], fx['node_obs'][t], adj_of(fx['adj'][t]), fx['reward'][t], fx['done'][t])
        for k in ('share_obs', 'obs', 'node_obs', 'agent_id', 'share_agent_id', 'rewards', 'masks', 'active_masks'):
            assert np.array_equal(getattr(buf, k), fx['ep%d_%s' % (ep, k)]), (ep, k)
        assert np.array_equal(buf.adj[:, :, 0], fx['ep%d_adj' % ep])
        buf.after_update()
        for k in ('share_obs', 'obs', 'node_obs', 'masks', 'active_masks'):
            assert np.array_equal(getattr(buf, k)[0], fx['ep%d_after_%s0' % (ep, k)]), (ep, k)
    assert (fx['ep0_masks'] == 0).any()


@pytest.mark.parametrize('name', RUNNER)
def test_process_adj_oracle_equals_reference(name):
    fx = load(name)
    args = json.loads(str(fx['args']))
    N = args['num_agents']
    found = 0
    for key in fx.files:
        if key.endswith('_index') and '_padj' in key:
            ep, s = int(key[2:key.index('_')]), int(key[key.index('padj') + 4:key.rindex('_')])
            adj = fx['ep%d_adj' % ep][s]                                     # (n, E, E) float32 buffer slot
            batch = np.repeat(adj[:, None], N, axis=1).reshape(-1, *adj.shape[1:])   # np.concatenate(buffer.adj[step])
            ei, ea = ro.process_adj(batch, args['max_edge_dist'])
            assert np.array_equal(ei, fx[key]) and np.array_equal(ea, fx[key.replace('_index', '_attr')])
            found += 1
    assert found >= 2


@pytest.mark.parametrize('name', RUNNER)
def test_update_graph_oracle_equals_reference(name):
    fx = load(name)
    args = json.loads(str(fx['args']))
    for t in range(fx['ent_pos'].shape[0]):
        for e in range(fx['ent_pos'].shape[1]):
            el, ew = ro.update_graph(fx['ent_pos'][t, e], args['max_edge_dist'])
            k = int(fx['edge_nnz'][t, e])
            assert el.shape[1] == k and np.array_equal(el, fx['edge_list'][t, e][:, :k])
            assert np.array_equal(ew, fx['edge_weight'][t, e][:k])   # bit-exact float64


def test_edge_threshold_known_answers():
    fx = load('edges_kat.npz')
    args = json.loads(str(fx['args']))
    for c in range(fx['ent_pos'].shape[0]):
        el, ew = ro.update_graph(fx['ent_pos'][c], args['max_edge_dist'])
        k = int(fx['edge_nnz'][c])
        assert el.shape[1] == k and np.array_equal(el, fx['edge_list'][c][:, :k]) and np.array_equal(ew, fx['edge_weight'][c][:k])
        pairs = set(map(tuple, el.T))
        assert (0, 1) in pairs and (0, 3) in pairs and (0, 2) not in pairs and (0, 4) not in pairs
        d = fx['ent_pos'][c][:, None] - fx['ent_pos'][c][None]
        adj32 = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32)
        ei, ea = ro.process_adj(adj32, args['max_edge_dist'])
        k = int(fx['padj_nnz'][c])
        assert ei.shape[1] == k and np.array_equal(ei, fx['padj_index'][c][:, :k])
        assert np.array_equal(ea.astype(np.float64), fx['padj_attr'][c][:k])
        assert (0, 1) not in set(map(tuple, ei.T))   # strict < : exactly max_edge_dist is no policy edge


@pytest.mark.parametrize('name', RUNNER)
def test_process_infos_oracle_equals_reference(name):
    fx = load(name)
    args = json.loads(str(fx['args']))
    T, N = args['episode_length'], args['num_agents']
    keys = [str(k) for k in fx['info_keys']]
    for ep in range(fx['actions'].shape[0] // T):
        info = fx['info'][(ep + 1) * T - 1]                # the infos of the episode's last step (graph_mpe_runner.py:146)
        got = ro.process_infos(info, keys, T)
        names = [str(s) for s in fx['ep%d_env_info_names' % ep]]
        want = fx['ep%d_env_infos' % ep]                   # (names, N, n), NaN rows = empty lists
        assert sorted(set(k.split('/', 1)[1] for k in got)) == names
        for j, nm in enumerate(names):
            for a in range(N):
                v = got['agent%d/%s' % (a, nm)]
                if len(v) == 0:
                    assert np.isnan(want[j, a]).all()
                else:
                    assert np.array_equal(np.array(v), want[j, a]), (nm, a)
        for reader in ro.METRIC_PATTERNS:
            key = 'ep%d_%s' % (ep, reader)
            if key in fx.files:
                assert np.array_equal(np.array(ro.metric(got, reader)), fx[key]), reader
            else:
                with pytest.raises(IndexError):
                    ro.metric(got, reader)
