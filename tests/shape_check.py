"""Digest of a rollout for tests/test_hip_parity.py::test_shape_instances_equal_the_generic_kernels: run by the test in this process
(the library's shape instances: step kernels with the agent / obstacle / wall counts as compile-time constants) and in a child with
FMARL_GENERIC_SHAPES=1 (the generic kernels); the digests must be equal -- same arithmetic in the same order.
usage: python tests/shape_check.py <case> -> one line "DIGEST <case> <sha256 of every step's outputs and the final state>"."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fair_marl_amd as fm  # noqa: E402

CASES = {
    # (env kwargs, n_envs): the shapes libfmarl has instances for (fmarl_step.hip kNavShapes, formation_shape_const, fairnav NL = 3)
    'nav3_small': (dict(num_agents=3, num_landmarks=3, num_obstacles=3), 4096),            # BASELINE config 2: the small-batch kernels
    'nav10': (dict(num_agents=10, num_landmarks=10, num_obstacles=3), 6000),               # the reference's own scale: step / step_end / span kernels
    'form10': (dict(scenario_name='fair_graph_formation', num_agents=10, num_landmarks=1, num_obstacles=3), 3000),   # BASELINE config 4
    'fnav3': (dict(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=3, num_landmarks=3, num_obstacles=3, goal_rew=30.0,
                   collision_rew=30.0, min_dist_thresh=0.4), 5000),                         # the shipped FA+FR shape, episodes ending at all phases
    'fnav10': (dict(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=10, num_landmarks=10, num_obstacles=3, goal_rew=30.0,
                    collision_rew=30.0, min_dist_thresh=0.3), 1500),                        # BASELINE.md section 2's N = 10 row (per-step kernel only)
}


def digest(case):
    kw, n = CASES[case]
    cfg = fm.EnvConfig(**dict(kw, episode_length=7))
    dev = torch.device('cuda:0')
    h = hashlib.sha256()
    for mode in ('eager', 'span'):
        eng = fm.RolloutEngine(cfg, n, device=dev, seed=77)
        ring = fm.OutputRing(eng, 16)
        g = torch.Generator(device=dev); g.manual_seed(5)
        tape = torch.randint(0, 5, (16, n, cfg.N), device=dev, generator=g, dtype=torch.int32)
        eng.reset()
        eng.rollout(tape, mode=mode, ring=ring)   # two episode ends inside (auto-resets: staged / in-kernel)
        torch.cuda.synchronize()
        for name in ('obs', 'reward', 'done', 'node_obs', 'adj_env', 'info_planes'):
            h.update(getattr(ring, name).cpu().numpy().tobytes())
        st = eng.get_state()
        for key in sorted(st):
            h.update(np.ascontiguousarray(st[key]).tobytes())
        eng.close()
    return h.hexdigest()


if __name__ == '__main__':
    print('DIGEST %s %s' % (sys.argv[1], digest(sys.argv[1])))
