"""Pins the oracle: replays every committed golden fixture (outputs of the reference itself,
see tests/golden/gen_golden.py) through oracle/nav_oracle.py.  CPU only."""
import numpy as np
import pytest

from oracle import nav_oracle as no
from helpers import TRAJ, cfg_of, load, state_from

TOL = dict(rtol=1e-9, atol=1e-9)  # float64 vs float64; N=32 contacts amplify rounding to ~1e-11


@pytest.mark.parametrize('name', TRAJ)
def test_trajectory_fixture(name):
    fx = load(name)
    cfg = cfg_of(fx)
    st = state_from(fx, cfg)
    for t in range(fx['actions'].shape[0]):
        out = no.env_step(cfg, st, fx['actions'][t])
        for k in ('obs', 'node_obs', 'adj', 'reward', 'info'):
            np.testing.assert_allclose(out[k], fx[k][t], err_msg='%s step %d %s' % (name, t, k), **TOL)
        assert np.array_equal(out['done'], fx['done'][t])
    for k in no.State.FIELDS:
        np.testing.assert_allclose(getattr(st, k), fx['final_' + k], err_msg=k, **TOL)


def test_cfg1_dummy_vec_env_bit_stream():
    """BASELINE config 1: GraphDummyVecEnv x 8, N=3, incl. two auto-resets, NumPy global stream."""
    fx = load('cfg1_dummy8.npz')
    cfg = cfg_of(fx)
    n, seed = fx['reset_obs'].shape[0], int(fx['seed'])
    np.random.seed(seed)
    env = no.OracleGraphVecEnv(cfg, n, seeds=[seed + 1000 * r for r in range(n)], mode='dummy')
    obs, ids, node, adj = env.reset()
    assert obs.dtype == np.float64 and ids.dtype == np.int64 and ids.shape == (n, cfg.N, 1)
    np.testing.assert_array_equal(obs, fx['reset_obs'])
    np.testing.assert_array_equal(ids, fx['reset_id'])
    np.testing.assert_array_equal(node, fx['reset_node_obs'])
    np.testing.assert_array_equal(adj[:, 0], fx['reset_adj'])
    assert adj.shape == (n, cfg.N, cfg.E, cfg.E)
    resets = 0
    for t in range(fx['actions'].shape[0]):
        res = env.step(fx['actions'][t])
        assert len(res) == 8
        o, i, nd, ad, r, d, info, rc = res
        np.testing.assert_allclose(o, fx['obs'][t], **TOL)
        np.testing.assert_allclose(nd, fx['node_obs'][t], **TOL)
        np.testing.assert_allclose(ad[:, 0], fx['adj'][t], **TOL)
        np.testing.assert_allclose(r, fx['reward'][t], **TOL)
        np.testing.assert_allclose(info, fx['info'][t], **TOL)
        assert np.array_equal(d, fx['done'][t]) and rc == fx['reset_count'][t]
        resets += rc
    assert resets == 2


def test_world_step_known_answers():
    """Single World.step() answers captured from the reference's core.py."""
    fx = load('kat_world.npz')
    for i in range(int(fx['n_kats'])):
        ag, ob, wl = fx['kat%d_agents' % i], fx['kat%d_obstacles' % i], fx['kat%d_walls' % i]
        cfg = no.Config(num_agents=len(ag), num_landmarks=0, num_obstacles=len(ob), num_walls=len(wl))
        st = no.State(cfg, 1)
        st.agent_pos[0] = ag[:, :2]; st.agent_vel[0] = ag[:, 2:]
        st.obstacle_pos[0] = ob
        if len(wl):
            st.wall_orient[0] = wl[:, 0]; st.wall_axis[0] = wl[:, 1]; st.wall_e0[0] = wl[:, 2]; st.wall_e1[0] = wl[:, 3]
        for row in fx['kat%d_out' % i]:
            no.world_step(cfg, st, fx['kat%d_u' % i][None])
            got = np.concatenate([st.agent_pos[0], st.agent_vel[0], st.p_dist[0][:, None]], axis=1)
            np.testing.assert_allclose(got, row, rtol=1e-12, atol=1e-14)


def test_survey_known_answers():
    """SURVEY.md App. B KAT 1, 2, 5, 6 as literal numbers."""
    cfg = no.Config(num_agents=1, num_landmarks=0, num_obstacles=0)
    st = no.State(cfg, 1)
    no.world_step(cfg, st, np.array([[[5.0, 0.0]]]))
    assert np.allclose(st.agent_pos[0, 0], [0.05, 0]) and np.allclose(st.agent_vel[0, 0], [0.5, 0])
    no.world_step(cfg, st, np.array([[[5.0, 0.0]]]))
    assert np.allclose(st.agent_pos[0, 0], [0.1375, 0]) and np.isclose(st.p_dist[0, 0], 0.1375)
    cfg = no.Config(num_agents=2, num_landmarks=0, num_obstacles=0)
    st = no.State(cfg, 1); st.agent_pos[0] = [[0, 0], [0.08, 0]]
    no.world_step(cfg, st, np.zeros((1, 2, 2)))
    assert np.allclose(st.agent_vel[0, :, 0], [-0.7879570125, 0.7879570125], atol=1e-10)
    assert np.isclose(st.p_dist[0, 0], 0.07879570125109339, atol=1e-15)
    for pd, f in (([0.05, 0.05, 0], 1.408238910673487), ([0.3, 0.31, 0.29], 36.29779081036919),
                  ([1, 0.2, 0.5], 1.716739075235204)):
        assert np.isclose(no.fairness_from(np.array(pd)), f, rtol=1e-13)
