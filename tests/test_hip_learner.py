"""-m gpu: the learner's side of the device rollout buffer through the C-ABI --
fmarl_compute_returns, fmarl_advantages, fmarl_minibatch_gather behind DeviceRolloutBuffer.compute_returns / advantages /
feed_forward_generator / recurrent_generator -- against the reference's own outputs (tests/golden/learner_*.npz, produced by
running GraphReplayBuffer / GR_MAPPO.train / ValueNorm / PopArt: gen_learner.py) and, at larger sizes, against the oracle
restatement that those fixtures pin (oracle/runner_oracle.py)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest
import torch

import fair_marl_amd as fm
from fair_marl_amd import _lib
from fair_marl_amd.rollout_buffer import DeviceRolloutBuffer
from oracle import runner_oracle as ro

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, 'golden', 'learner_*.npz')))
KEYS = DeviceRolloutBuffer.GENERATOR_FIELDS


def filled_buffer(arrays, T, n, N, E, R, H, landmarks=None):
    """A DeviceRolloutBuffer of the given shape whose tensors hold ``arrays`` (name -> ndarray in the reference's shapes)."""
    L = landmarks if landmarks is not None else N
    cfg = fm.EnvConfig(num_agents=N, num_landmarks=L, num_obstacles=E - N - L, episode_length=T)
    assert cfg.E == E
    eng = fm.RolloutEngine(cfg, n, device=DEV, seed=1)
    buf = DeviceRolloutBuffer(eng).attach_policy(act_dim=1, recurrent_N=R, hidden_size=H, n_actions=5)
    for k, v in arrays.items():
        getattr(buf, k).copy_(torch.as_tensor(v, device=DEV).reshape(getattr(buf, k).shape))
    return buf


def fixture_arrays(z):
    return dict(obs=z['buf_obs'], node_obs=z['buf_node_obs'], adj_env=z['buf_adj_env'], rewards=z['buf_rewards'], masks=z['buf_masks'],
                active_masks=z['buf_active_masks'], bad_masks=z['buf_bad_masks'], value_preds=z['buf_value_preds'], actions=z['buf_actions'],
                action_log_probs=z['buf_action_log_probs'], rnn_states=z['buf_rnn_states'], rnn_states_critic=z['buf_rnn_states_critic'],
                available_actions=z['buf_available_actions'])


@pytest.mark.parametrize('path', FIXTURES, ids=os.path.basename)
def test_learner_side_equals_the_reference(path):
    z = np.load(path)
    T, n, N, D, E, F, H, R, nmb, L = [int(x) for x in z['shape']]
    buf = filled_buffer(fixture_arrays(z), T, n, N, E, R, H)
    assert (buf.engine.cfg.obs_dim, buf.engine.cfg.node_feat) == (D, F)
    nv = torch.as_tensor(z['next_value'], device=DEV)
    gamma, lam = float(z['gamma']), float(z['gae_lambda'])
    # compute_returns: all twelve branches, bit for bit
    for gae in (1, 0):
        for proper in (0, 1):
            for nm in ('none', 'valuenorm', 'popart'):
                norm = None if nm == 'none' else tuple(float(x) for x in z['norm_' + nm])
                buf.returns.zero_()
                buf.value_preds.copy_(torch.as_tensor(z['buf_value_preds'], device=DEV))
                got = buf.compute_returns(nv, norm, gamma, lam, use_gae=bool(gae), use_proper_time_limits=bool(proper)).cpu().numpy()
                want = z['ret_%d%d_%s' % (gae, proper, nm)]
                assert np.array_equal(got, want), (gae, proper, nm, float(np.abs(got - want).max()))
                if gae:
                    assert np.array_equal(buf.value_preds[-1].cpu().numpy(), z['next_value'])
    # advantages (graph_mappo.py:294-304): float64 accumulation here, float32 pairwise sums there
    for nm in ('valuenorm', 'none'):
        norm = None if nm == 'none' else tuple(float(x) for x in z['norm_' + nm])
        buf.compute_returns(nv, norm, gamma, lam)
        adv = buf.advantages(norm)
        np.testing.assert_allclose(adv.cpu().numpy(), z['adv_' + nm], rtol=2e-6, atol=2e-6)
        again = buf.advantages(norm)
        assert torch.equal(adv, again)   # fixed summation order: the same bits on every call
    assert np.array_equal(buf.returns.cpu().numpy(), z['gen_returns'])
    adv_ref = torch.as_tensor(z['adv_none'], device=DEV)
    # the two generators with the reference's permutations: all 16 arrays of every minibatch, bit for bit
    for tag, gen in (('ff', buf.feed_forward_generator(adv_ref, nmb, perm=torch.as_tensor(z['ff_perm']))),
                     ('rec', buf.recurrent_generator(adv_ref, nmb, L, perm=torch.as_tensor(z['rec_perm'])))):
        count = 0
        for b, sample in enumerate(gen):
            assert len(sample) == 16
            for k, g in zip(KEYS, sample):
                want = z['%s%d_%s' % (tag, b, k)]
                assert g.dtype == (torch.int32 if 'agent_id' in k else torch.float32) and tuple(g.shape) == want.shape, (tag, b, k, g.shape)
                assert np.array_equal(g.cpu().numpy(), want), (tag, b, k)
            count += 1
        assert count == nmb


@pytest.mark.parametrize('shape', [dict(T=25, n=301, N=10, L=10, O=3, R=1, H=16), dict(T=9, n=64, N=32, L=32, O=8, R=2, H=8),
                                   dict(T=5, n=7, N=1, L=1, O=0, R=1, H=4)], ids=lambda s: 'n%d-N%d' % (s['n'], s['N']))
def test_learner_side_equals_the_oracle_at_larger_sizes(shape):
    T, n, N, R, H = shape['T'], shape['n'], shape['N'], shape['R'], shape['H']
    E = N + shape['L'] + shape['O']
    cfg = fm.EnvConfig(num_agents=N, num_landmarks=shape['L'], num_obstacles=shape['O'], episode_length=T)
    D, F = cfg.obs_dim, cfg.node_feat
    rs = np.random.RandomState(T * n + N)
    f32 = np.float32
    done = rs.rand(T + 1, n, N, 1) < 0.1
    arrays = dict(obs=rs.randn(T + 1, n, N, D).astype(f32), node_obs=rs.randn(T + 1, n, N, E, F).astype(f32),
                  adj_env=rs.rand(T + 1, n, E, E).astype(f32), rewards=(4 * rs.randn(T, n, N, 1)).astype(f32),
                  masks=(~done).astype(f32), active_masks=(rs.rand(T + 1, n, N, 1) > 0.2).astype(f32),
                  bad_masks=(rs.rand(T + 1, n, N, 1) > 0.1).astype(f32), value_preds=rs.randn(T + 1, n, N, 1).astype(f32),
                  actions=rs.randint(0, 5, (T, n, N, 1)).astype(f32), action_log_probs=rs.randn(T, n, N, 1).astype(f32),
                  rnn_states=rs.randn(T + 1, n, N, R, H).astype(f32), rnn_states_critic=rs.randn(T + 1, n, N, R, H).astype(f32),
                  available_actions=(rs.rand(T + 1, n, N, 5) > 0.3).astype(f32))
    buf = filled_buffer(arrays, T, n, N, E, R, H, landmarks=shape['L'])
    nv = rs.randn(n, N, 1).astype(f32)
    norm = (f32(0.37), f32(1.9))
    for gae, proper, nrm in ((1, 0, norm), (1, 1, norm), (0, 1, None), (1, 1, None), (0, 0, norm)):
        buf.value_preds.copy_(torch.as_tensor(arrays['value_preds'], device=DEV))
        buf.returns.zero_()
        got = buf.compute_returns(torch.as_tensor(nv, device=DEV), nrm, 0.99, 0.95, bool(gae), bool(proper)).cpu().numpy()
        want, v = ro.compute_returns(arrays['rewards'], arrays['value_preds'], arrays['masks'], arrays['bad_masks'], nv, 0.99, 0.95,
                                     bool(gae), bool(proper), nrm)
        assert np.array_equal(got, want), (gae, proper, nrm)
    # (the last call: discounted sums, ValueNorm) -> advantages, generators
    adv = buf.advantages(norm)
    want_adv = ro.advantages(want, arrays['value_preds'], arrays['active_masks'], norm)
    np.testing.assert_allclose(adv.cpu().numpy(), want_adv, rtol=1e-5, atol=1e-5)
    stats = buf.advantage_stats.cpu().numpy()
    raw = (want[:-1] - (arrays['value_preds'][:-1] * norm[1] + norm[0]))[arrays['active_masks'][:-1] != 0].astype(np.float64)
    np.testing.assert_allclose(stats, [raw.mean(), raw.std()], rtol=1e-6)
    host = dict(arrays, returns=want, agent_id=np.broadcast_to(np.arange(N, dtype=np.int32).reshape(1, 1, N, 1), (T + 1, n, N, 1)),
                share_agent_id=np.broadcast_to(np.arange(N, dtype=np.int32).reshape(1, 1, 1, N), (T + 1, n, N, N)))
    advh = adv.cpu().numpy()
    nmb, L = 3, 10 if T >= 10 else 3
    perm = rs.permutation(T * n * N)
    for b, (sample, rows) in enumerate(zip(buf.feed_forward_generator(adv, nmb, perm=torch.as_tensor(perm), with_env_slot=True),
                                           ro.feed_forward_rows(perm, T, n, N, nmb))):
        ref = ro.gather_minibatch(host, advh, rows)
        for k, g, w in zip(KEYS, sample, ref):
            assert np.array_equal(g.cpu().numpy(), w), ('ff', b, k)
        assert np.array_equal(sample[16].cpu().numpy(), rows[0] * n + rows[1])
        # what the env_slot is for: the per-env adj indexed instead of materialised
        assert torch.equal(buf.adj_env.view(-1, E, E)[sample[16]], sample[3])
    assert b == nmb - 1
    cperm = rs.permutation(T * n * N // L)
    for b, (sample, (rows, first)) in enumerate(zip(buf.recurrent_generator(adv, nmb, L, perm=torch.as_tensor(cperm)),
                                                    ro.recurrent_rows(cperm, T, n, N, nmb, L))):
        ref = ro.gather_minibatch(host, advh, rows, first)
        for k, g, w in zip(KEYS, sample, ref):
            assert np.array_equal(g.cpu().numpy(), w), ('rec', b, k)
    # a subset of the fields: the others are not materialised
    sample = next(buf.feed_forward_generator(None, nmb, perm=torch.as_tensor(perm), fields=('obs', 'returns')))
    assert [k for k, g in zip(KEYS, sample) if g is not None] == ['obs', 'returns']
    # after_update carries the policy-side slots too (graph_buffer.py:253-283)
    buf.after_update()
    for k in ('rnn_states', 'rnn_states_critic', 'bad_masks', 'available_actions', 'masks', 'active_masks', 'obs'):
        assert torch.equal(getattr(buf, k)[0], getattr(buf, k)[-1]), k


def test_learner_entry_points_refuse_bad_arguments():
    lib = _lib.load()
    x = torch.zeros(64, device=DEV)
    args = _lib.FmarlReturns(0.99, 0.95, 0.0, 1.0, 0, 1, 1, 3, 4)
    p = x.data_ptr()
    assert lib.fmarl_compute_returns(C.byref(args), p, p, p, None, p, p, None) != 0
    assert b'bad_masks' in lib.fmarl_last_error()
    args.T = 0
    assert lib.fmarl_compute_returns(C.byref(args), p, p, p, p, p, p, None) != 0
    src = _lib.FmarlBatchSrc(*([None] * 13), 3, 4, 2, 6, 6, 7, 8, 1, 5, 0)
    dst = _lib.FmarlBatchDst(*([None] * 17))
    dst.obs = p
    idx = torch.zeros(8, dtype=torch.int64, device=DEV)
    assert lib.fmarl_minibatch_gather(C.byref(src), C.byref(dst), idx.data_ptr(), 8, 0, 1, None) != 0
    assert b'obs needs its source' in lib.fmarl_last_error()
    src.obs = p
    assert lib.fmarl_minibatch_gather(C.byref(src), C.byref(dst), idx.data_ptr(), 7, 1, 2, None) != 0
    assert b'multiple of the chunk' in lib.fmarl_last_error()
    assert lib.fmarl_minibatch_gather(C.byref(src), C.byref(dst), idx.data_ptr(), 8, 2, 1, None) != 0
    assert lib.fmarl_advantages(p, p, p, p, 0, 0, 0.0, 1.0, p, None) != 0
    buf = DeviceRolloutBuffer(fm.RolloutEngine(fm.EnvConfig(num_agents=3, num_landmarks=3, num_obstacles=2, episode_length=4), 3, device=DEV))
    with pytest.raises(RuntimeError, match='attach_policy'):
        buf.compute_returns(torch.zeros(3, 3, 1))


def test_example_rollout_to_minibatches_runs():
    """examples/rollout_to_minibatches.py: rollout -> returns -> advantages -> recurrent minibatches on the device, tiny sizes;
    every (env, agent, step) appears in exactly one minibatch row (chunks partition the (n, N, T) series)."""
    import importlib.util
    path = os.path.join(os.path.dirname(HERE), 'examples', 'rollout_to_minibatches.py')
    spec = importlib.util.spec_from_file_location('rollout_to_minibatches', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, N = 40, 3
    seen, buf = mod.main(n_envs=n, num_agents=N, iterations=2, num_mini_batch=2, data_chunk_length=10, hidden_size=8, device=DEV, verbose=False)
    assert seen == 2 * buf.T * n * N
    assert torch.isfinite(buf.returns).all() and float(buf.masks.min()) == 0.0   # an episode end fell inside the rollouts


def test_data_parallel_advantages_equal_the_unsharded_buffer():
    """Three ranks, one buffer shard each (ranks share the GPU, exchange over gloo): advantages(group=...) standardises with the
    statistics of the whole batch -- equal to one buffer over all envs (tests/dist_learner_check.py)."""
    from test_hip_parity import _run_ranks
    out = _run_ranks([os.path.join(HERE, 'dist_learner_check.py')], world=3)
    assert 'DIST_LEARNER_OK world=3' in out, out[-4000:]
