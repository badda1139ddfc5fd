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


@pytest.mark.parametrize('name', RUNNER)
def test_engine_equals_the_reference_on_the_philox_stream(name):
    """No state injection anywhere: device resets (placement, assignment, reset observation) and steps against the
    reference's own outputs through every auto-reset of the run."""
    fx = load(name)
    eng, cfg, args, n = engine_of(fx)
    obs, ids, node, adj = eng.reset()
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
