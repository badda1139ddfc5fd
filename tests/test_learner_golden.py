"""CPU: the oracle's restatement of the learner side of the buffer (oracle/runner_oracle.py: compute_returns, advantages,
the two minibatch generators) against the reference's own outputs (tests/golden/learner_*.npz, gen_learner.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import runner_oracle as ro

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, 'golden', 'learner_*.npz')))
GEN_KEYS = ('share_obs', 'obs', 'node_obs', 'adj', 'agent_id', 'share_agent_id', 'rnn_states', 'rnn_states_critic', 'actions',
            'value_preds', 'returns', 'masks', 'active_masks', 'old_action_log_probs', 'adv_targ', 'available_actions')


def load(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def norm_of(z, nm):
    return None if nm == 'none' else tuple(z['norm_' + nm])


@pytest.mark.parametrize('path', FIXTURES, ids=os.path.basename)
def test_compute_returns_every_branch_bit_exact(path):
    z = load(path)
    assert len(FIXTURES) >= 2
    for gae in (1, 0):
        for proper in (0, 1):
            for nm in ('none', 'valuenorm', 'popart'):
                ret, v = ro.compute_returns(z['buf_rewards'], z['buf_value_preds'], z['buf_masks'], z['buf_bad_masks'], z['next_value'],
                                            float(z['gamma']), float(z['gae_lambda']), bool(gae), bool(proper), norm_of(z, nm))
                want = z['ret_%d%d_%s' % (gae, proper, nm)]
                assert np.array_equal(ret, want), (gae, proper, nm, np.abs(ret - want).max())
                if gae:
                    assert np.array_equal(v[-1], z['next_value'])


@pytest.mark.parametrize('path', FIXTURES, ids=os.path.basename)
def test_advantages(path):
    z = load(path)
    for nm in ('valuenorm', 'none'):
        ret, v = ro.compute_returns(z['buf_rewards'], z['buf_value_preds'], z['buf_masks'], z['buf_bad_masks'], z['next_value'],
                                    float(z['gamma']), float(z['gae_lambda']), True, False, norm_of(z, nm))
        adv = ro.advantages(ret, v, z['buf_active_masks'], norm_of(z, nm))
        np.testing.assert_allclose(adv, z['adv_' + nm], rtol=2e-6, atol=2e-6)   # float32 mean / std: summation order
        assert (z['buf_active_masks'][:-1] == 0).any()


@pytest.mark.parametrize('path', FIXTURES, ids=os.path.basename)
def test_generators_bit_exact(path):
    z = load(path)
    T, n, N, D, E, F, H, R, nmb, L = [int(x) for x in z['shape']]
    buf = {k[4:]: v for k, v in z.items() if k.startswith('buf_')}
    buf['value_preds'] = buf['value_preds'].copy()
    buf['value_preds'][-1] = z['next_value']
    buf['returns'] = z['gen_returns']
    adv = z['adv_none']
    for b, rows in enumerate(ro.feed_forward_rows(z['ff_perm'], T, n, N, nmb)):
        got = ro.gather_minibatch(buf, adv, rows)
        for k, g in zip(GEN_KEYS, got):
            assert np.array_equal(g, z['ff%d_%s' % (b, k)]), ('ff', b, k)
    for b, (rows, first) in enumerate(ro.recurrent_rows(z['rec_perm'], T, n, N, nmb, L)):
        got = ro.gather_minibatch(buf, adv, rows, first)
        for k, g in zip(GEN_KEYS, got):
            assert np.array_equal(g, z['rec%d_%s' % (b, k)]), ('rec', b, k)
    assert 'rec%d_obs' % nmb not in z and 'ff%d_obs' % nmb not in z
