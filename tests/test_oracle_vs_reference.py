"""Live check of the oracle against the imported reference (build container only; skipped
where /root/reference is absent, e.g. on the GPU box).  CPU only."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import refharness as rh  # noqa: E402
from oracle import nav_oracle as no  # noqa: E402

pytestmark = pytest.mark.skipif(not rh.available(), reason='reference not present')


@pytest.mark.parametrize('N,O,W,seed', [(3, 3, 0, 3), (3, 3, 2, 4), (5, 2, 1, 5), (10, 3, 0, 6)])
def test_random_rollout_matches_reference(N, O, W, seed):
    args = rh.make_args(num_agents=N, num_landmarks=N, num_obstacles=O, num_walls=W)
    np.random.seed(seed)
    env = rh.make_env(args); env.seed(seed); env.reset()
    cfg = no.Config.from_args(args)
    st = no.State(cfg, 1)
    s = rh.capture_state(env)
    for k in no.State.FIELDS:
        getattr(st, k)[0] = s[k]
    rs = np.random.RandomState(seed + 7)
    for t in range(30):
        a = rs.randint(0, 5, size=N)
        obs, ids, node, adj, rew, done, info = env.step([rh.onehot(x) for x in a])
        out = no.env_step(cfg, st, a[None])
        np.testing.assert_allclose(out['obs'][0], np.array(obs), rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(out['node_obs'][0], np.array(node), rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(out['adj'][0], adj[0], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(out['reward'][0], np.array(rew), rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(out['info'][0], rh.info_array(info, no.INFO_KEYS), rtol=1e-9, atol=1e-10)
        assert list(out['done'][0]) == list(done)
