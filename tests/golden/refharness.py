"""Build-container-only harness that imports the Python reference from ``/root/reference``.

Used by ``gen_golden.py`` (fixture generation) and by ``tests/test_oracle_vs_reference.py``
(skipped wherever ``/root/reference`` is absent, e.g. on the GPU box).  The reference needs
three third-party modules that are not installed here; they are stubbed in ``sys.modules``
(SURVEY.md App. C): ``gym`` (spaces only), ``absl`` (flags), and ``marl_fair_assign``
(pyomo + Gurobi) -- the latter replaced by the oracle's lexifair solver, which is why the
assignment itself is "parity unpinned" (see ``oracle/lexifair.py``).
"""
import argparse
import os
import sys
import types

import numpy as np

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def available():
    return os.path.isdir(os.path.join(REF, 'multiagent'))


def install_stubs():
    if 'gym' in sys.modules and getattr(sys.modules['gym'], '_fmarl_stub', False):
        return
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import lexifair as _lf

    gym = types.ModuleType('gym'); gym._fmarl_stub = True
    spaces = types.ModuleType('gym.spaces')
    reg = types.ModuleType('gym.envs.registration')
    envs = types.ModuleType('gym.envs')

    class Env(object):
        def close(self):
            pass

    class Space(object):
        pass

    class Box(Space):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    class Discrete(Space):
        def __init__(self, n):
            self.n = n
            self.shape = ()

    class Tuple_(Space):
        def __init__(self, spaces_):
            self.spaces = spaces_

    gym.Env = Env; gym.Space = Space; gym.spaces = spaces; gym.envs = envs
    spaces.Box = Box; spaces.Discrete = Discrete; spaces.Tuple = Tuple_; spaces.Space = Space
    reg.register = lambda **k: None
    envs.registration = reg
    sys.modules.update({'gym': gym, 'gym.spaces': spaces, 'gym.envs': envs, 'gym.envs.registration': reg})

    mfa = types.ModuleType('marl_fair_assign')
    mfa.solve_fair_assignment = _lf.solve_fair_assignment
    sys.modules['marl_fair_assign'] = mfa

    absl = types.ModuleType('absl'); flags = types.ModuleType('absl.flags')
    flags.FLAGS = lambda *a, **k: None
    absl.flags = flags
    sys.modules.update({'absl': absl, 'absl.flags': flags})
    if REF not in sys.path:
        sys.path.insert(0, REF)


def make_args(**kw):
    d = dict(scenario_name='navigation_graph', world_size=2, num_agents=3, num_scripted_agents=0,
             num_obstacles=3, collaborative=False, max_speed=2, collision_rew=5, goal_rew=5,
             min_dist_thresh=0.05, use_dones=False, episode_length=25, fair_wt=1, fair_rew=1,
             max_edge_dist=1, graph_feat_type='relative', num_landmarks=3, num_walls=0, zeroshift=5,
             algorithm_name='rmappo', env_name='GraphMPE', min_obs_dist=0.5)
    d.update(kw)
    return argparse.Namespace(**d)


def make_env(args):
    install_stubs()
    from multiagent.MPE_env import GraphMPEEnv
    return GraphMPEEnv(args)


def scenario_of(env):
    """The Scenario instance behind the env's bound callbacks."""
    return env.reset_callback.__self__


def onehot(idx, k=5):
    a = np.zeros(k)
    a[int(idx)] = 1.0
    return a


def capture_state(env):
    """Full per-world state in the oracle's ``State`` field naming (single env, no batch axis)."""
    w = env.world
    sc = scenario_of(env)
    N = len(w.agents)
    s = dict(
        agent_pos=np.array([a.state.p_pos for a in w.agents], dtype=np.float64).reshape(N, 2),
        agent_vel=np.array([a.state.p_vel for a in w.agents], dtype=np.float64).reshape(N, 2),
        p_dist=np.array([a.state.p_dist for a in w.agents], dtype=np.float64),
        time=np.array([a.state.time for a in w.agents], dtype=np.float64),
        landmark_pos=np.array([l.state.p_pos for l in w.landmarks], dtype=np.float64).reshape(len(w.landmarks), 2),
        obstacle_pos=np.array([o.state.p_pos for o in w.obstacles], dtype=np.float64).reshape(len(w.obstacles), 2),
        wall_axis=np.array([x.axis_pos for x in w.walls], dtype=np.float64),
        wall_e0=np.array([x.endpoints[0] for x in w.walls], dtype=np.float64),
        wall_e1=np.array([x.endpoints[1] for x in w.walls], dtype=np.float64),
        wall_orient=np.array([0 if x.orient == 'H' else 1 for x in w.walls], dtype=np.int64),
        wall_length=np.float64(sc.wall_length),
        goal_match=np.array(sc.goal_match_index, dtype=np.int64),
        dists_to_goal=np.array(w.dists_to_goal, dtype=np.float64),
        times_required=np.array(w.times_required, dtype=np.float64),
        dist_left=np.array(w.dist_left_to_goal, dtype=np.float64),
        num_obst_coll=np.array(w.num_obstacle_collisions, dtype=np.float64),
        num_agent_coll=np.array(w.num_agent_collisions, dtype=np.float64),
        min_time=np.array([a.goal_min_time for a in w.agents], dtype=np.float64),
        cur_step=np.int64(env.current_step),
    )
    if hasattr(sc, 'landmark_poses_occupied'):  # nav_fairassign_fairrew_formation_graph.py scenario-level state
        s.update(goal_occ=np.array(sc.landmark_poses_occupied, dtype=np.float64),
                 goal_history=np.array(sc.goal_history, dtype=np.float64),
                 goal_reached=np.array(sc.goal_reached, dtype=np.float64),
                 status=np.array([1.0 if a.status == True else 0.0 for a in w.agents]))  # noqa: E712
    if hasattr(sc, 'expected_poses'):  # fair_graph_formation.py scenario-level state
        s.update(slot_pos=np.array(sc.expected_poses, dtype=np.float64).reshape(N, 2),
                 slot_occ=np.array(sc.expected_poses_occupied, dtype=np.float64),
                 slot_delta=np.array(getattr(sc, 'delta_dists', np.zeros(N)), dtype=np.float64),
                 formation_done=np.array(w.formation_complete, dtype=np.float64))
    return s


def inject_state(env, s):
    """Overwrite the reference world with a captured/crafted state."""
    w = env.world
    sc = scenario_of(env)
    for i, a in enumerate(w.agents):
        a.state.p_pos = np.array(s['agent_pos'][i], dtype=np.float64)
        a.state.p_vel = np.array(s['agent_vel'][i], dtype=np.float64)
        a.state.p_dist = float(s['p_dist'][i])
        a.state.time = float(s['time'][i])
        a.goal_min_time = float(s['min_time'][i])
    for i, l in enumerate(w.landmarks):
        l.state.p_pos = np.array(s['landmark_pos'][i], dtype=np.float64)
    for i, o in enumerate(w.obstacles):
        o.state.p_pos = np.array(s['obstacle_pos'][i], dtype=np.float64)
    for i, x in enumerate(w.walls):
        x.axis_pos = float(s['wall_axis'][i])
        x.endpoints = np.array([s['wall_e0'][i], s['wall_e1'][i]], dtype=np.float64)
        x.orient = 'H' if int(s['wall_orient'][i]) == 0 else 'V'
        x.state.p_pos = np.array([0.0, x.axis_pos]) if x.orient == 'H' else np.array([x.axis_pos, 0.0])
    sc.goal_match_index = np.array(s['goal_match'], dtype=np.int64)
    w.dists_to_goal = np.array(s['dists_to_goal'], dtype=np.float64)
    w.times_required = np.array(s['times_required'], dtype=np.float64)
    w.dist_left_to_goal = np.array(s['dist_left'], dtype=np.float64)
    w.num_obstacle_collisions = np.array(s['num_obst_coll'], dtype=np.float64)
    w.num_agent_collisions = np.array(s['num_agent_coll'], dtype=np.float64)
    # the stale statistics the next agent-0 obs/reward would read (navigation_graph.py:617-621)
    w.dist_traveled_mean = np.mean(w.dists_to_goal)
    w.dist_traveled_stddev = np.std(w.dists_to_goal)
    w.time_taken_mean = np.mean(w.times_required)
    w.time_taken_stddev = np.std(w.times_required)
    env.current_step = int(s['cur_step'])
    w.current_time_step = int(s['cur_step'])
    if 'goal_occ' in s and hasattr(sc, 'landmark_poses_occupied'):
        sc.landmark_poses = np.array(s['landmark_pos'], dtype=np.float64)
        sc.landmark_poses_occupied = np.array(s['goal_occ'], dtype=np.float64)
        sc.goal_history = np.array(s['goal_history'], dtype=np.float64)
        sc.goal_reached = np.array(s['goal_reached'], dtype=np.float64)
        for a, v in zip(w.agents, s['status']):
            a.status = bool(v)
    if 'slot_pos' in s and hasattr(sc, 'expected_poses'):
        sc.expected_poses = np.array(s['slot_pos'], dtype=np.float64)
        sc.expected_poses_occupied = np.array(s['slot_occ'], dtype=np.float64)
        sc.delta_dists = np.array(s['slot_delta'], dtype=np.float64)
        w.formation_complete = np.array(s['formation_done'], dtype=np.float64)
    w.calculate_distances()


def info_array(info_n, keys):
    return np.array([[float(d[k]) for k in keys] for d in info_n], dtype=np.float64)
