"""Pins the nav_fairassign_fairrew_formation_graph oracle against reference fixtures."""
import numpy as np
import pytest

from oracle import fairnav_oracle as fnv
from helpers import FNAV, fnav_cfg_of, fnav_state_from, load

TOL = dict(rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize('name', FNAV)
def test_fairnav_trajectory_fixture(name):
    fx = load(name)
    cfg = fnav_cfg_of(fx)
    st = fnav_state_from(fx, cfg)
    for t in range(fx['actions'].shape[0]):
        out = fnv.env_step(cfg, st, fx['actions'][t])
        for k in ('obs', 'node_obs', 'adj', 'reward', 'info'):
            np.testing.assert_allclose(out[k], fx[k][t], err_msg='%s step %d %s' % (name, t, k), **TOL)
        assert np.array_equal(out['done'], fx['done'][t])
    for k in fnv.State.FIELDS:
        np.testing.assert_allclose(getattr(st, k), fx['final_' + k], err_msg=k, **TOL)


def test_fairnav_dummy_vec_env_with_early_resets():
    fx = load('fnav_dummy3.npz')
    cfg = fnav_cfg_of(fx)
    n, seed = fx['reset_obs'].shape[0], int(fx['seed'])
    np.random.seed(seed)
    env = fnv.OracleFairNavVecEnv(cfg, n, seeds=[seed + 1000 * r for r in range(n)], mode='dummy')
    obs, ids, node, adj = env.reset()
    np.testing.assert_array_equal(obs, fx['reset_obs'])
    np.testing.assert_array_equal(node, fx['reset_node_obs'])
    assert obs.shape == (n, cfg.N, 11) and node.shape == (n, cfg.N, cfg.E, 13)
    resets = 0
    for t in range(fx['actions'].shape[0]):
        o, i, nd, ad, r, d, info, rc = env.step(fx['actions'][t])
        np.testing.assert_allclose(o, fx['obs'][t], **TOL)
        np.testing.assert_allclose(nd, fx['node_obs'][t], **TOL)
        np.testing.assert_allclose(ad[:, 0], fx['adj'][t], **TOL)
        np.testing.assert_allclose(r, fx['reward'][t], **TOL)
        np.testing.assert_allclose(info, fx['info'][t], **TOL)
        assert np.array_equal(d, fx['done'][t]) and rc == fx['reset_count'][t]
        resets += rc
    assert resets == int(fx["reset_count"].sum()) and resets >= 2
