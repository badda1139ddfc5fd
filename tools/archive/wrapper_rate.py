#!/usr/bin/env python
"""Measurement aid: agent-steps/s THROUGH the NumPy-compatible wrappers (device -> host copies, float64
conversion, (n, N, E, E) adj), i.e. the PCIe-inclusive rate of the drop-in boundary.  Not bench.py's `value`."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fair_marl_amd as fm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--n-envs', type=int, default=4096)
ap.add_argument('--agents', type=int, default=3)
ap.add_argument('--obstacles', type=int, default=3)
ap.add_argument('--steps', type=int, default=50)
a = ap.parse_args()
args = argparse.Namespace(scenario_name='navigation_graph', num_agents=a.agents, num_landmarks=a.agents,
                          num_obstacles=a.obstacles, num_walls=0, episode_length=25)


def get_env_fn(rank):
    def init():
        env = fm.GraphMPEEnv(args)
        env.seed(1 + rank * 1000)
        return env
    return init


venv = fm.GraphSubprocVecEnv([get_env_fn(i) for i in range(a.n_envs)])
venv.reset()
rs = np.random.RandomState(0)
acts = np.eye(5)[rs.randint(0, 5, size=(a.n_envs, a.agents))]
for _ in range(3):
    venv.step(acts)
t0 = time.perf_counter()
for _ in range(a.steps):
    out = venv.step(acts)
dt = time.perf_counter() - t0
host_bytes = sum(x.nbytes for x in out[:6])
print('wrappers: n_envs=%d N=%d  %.2f ms/step  %.3e agent-steps/s  (%.1f MB of NumPy outputs per step)'
      % (a.n_envs, a.agents, dt / a.steps * 1e3, a.n_envs * a.agents * a.steps / dt, host_bytes / 1e6))
