"""measurement aid (GPU box): agent-steps/s THROUGH the drop-in NumPy API (GraphSubprocVecEnv.step: host actions in, float64 arrays out --
PCIe and the host's copies included; never bench.py's `value`).  usage: python tools/wrapper_rate.py [envs=4096] [agents=3] [steps=200]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fair_marl_amd as fm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200


class Args:
    scenario_name = 'navigation_graph'; num_agents = N; num_landmarks = N; num_obstacles = 3 if N < 20 else 8; num_walls = 0
    world_size = 2; max_speed = 2; collision_rew = 5; goal_rew = 5; min_dist_thresh = 0.05; fair_wt = 1; fair_rew = 1; zeroshift = 5
    max_edge_dist = 1; episode_length = 25; collaborative = False; use_dones = False; graph_feat_type = 'relative'; num_scripted_agents = 0
    min_obs_dist = 0.5


def fn(rank):
    def init():
        env = fm.GraphMPEEnv(Args())
        env.seed(1 + rank * 1000)
        return env
    return init


venv = fm.GraphSubprocVecEnv([fn(i) for i in range(n)])
venv.reset()
rs = np.random.RandomState(0)
acts = [rs.randint(0, 5, size=(n, N)) for _ in range(8)]
for k in range(10):
    venv.step(acts[k % 8])
t0 = time.perf_counter()
for k in range(steps):
    out = venv.step(acts[k % 8])
dt = time.perf_counter() - t0
print('%d envs x %d agents through GraphSubprocVecEnv.step: %.3f ms per step, %.3g agent-steps/s (FMARL_FETCH_F64=%s); obs %s %s node_obs %s'
      % (n, N, dt / steps * 1e3, n * N * steps / dt, os.environ.get('FMARL_FETCH_F64', '0'), out[0].dtype, out[0].shape, out[2].shape))
