#!/usr/bin/env python
"""Generate the committed golden fixtures by RUNNING THE REFERENCE (build container only).

    python tests/golden/gen_golden.py

imports ``/root/reference`` through ``refharness`` (gym / absl / Gurobi stubbed) and writes
``tests/golden/*.npz``: inputs (initial world state + action tape) and the reference's own
outputs per step.  The fixtures are data only; no reference source travels.

Fixture families
  cfg1_dummy8.npz      BASELINE config 1: the reference's GraphDummyVecEnv x 8 envs, N=3, seed 1,
                       60 random steps incl. two auto-resets, NumPy global RNG stream.
  traj_<case>.npz      state-injected trajectories of bare MultiAgentGraphEnv worlds: the state
                       after a seeded reference reset (or a crafted one), an action tape, and
                       per-step obs / node_obs / adj / reward / done / info + the final state.
  kat_world.npz        single World.step() known answers (SURVEY.md App. B KAT 1-5).
  fnav_<case>.npz      the same for nav_fairassign_fairrew_formation_graph.
  form_<case>.npz      the same for fair_graph_formation (BASELINE config 4), form_dummy4.npz its
                       GraphDummyVecEnv x 4 run incl. auto-resets.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import refharness as rh  # noqa: E402
from oracle.nav_oracle import INFO_KEYS, State  # noqa: E402
from oracle import formation_oracle as fo  # noqa: E402
from oracle import fairnav_oracle as fnv  # noqa: E402

rh.install_stubs()

STATE_FIELDS = State.FIELDS


def stack_states(states, fields=STATE_FIELDS):
    return {k: np.stack([np.asarray(s[k]) for s in states]) for k in fields}


def run_traj(args, states_or_seeds, actions, crafted=None, info_keys=INFO_KEYS, fields=STATE_FIELDS):
    """actions: (T, n, N) int or (T, n, N, 5) float.  Returns dict of arrays."""
    n = len(states_or_seeds)
    T = actions.shape[0]
    envs, init = [], []
    for e in range(n):
        np.random.seed(1000 + e)
        env = rh.make_env(args)
        if crafted is None:
            env.seed(int(states_or_seeds[e]))
            env.reset()
        else:
            env.seed(0)
            env.reset()
            s = rh.capture_state(env)
            s.update(states_or_seeds[e])
            rh.inject_state(env, s)
        envs.append(env)
        init.append(rh.capture_state(env))
    out = {k: [] for k in ('obs', 'node_obs', 'adj', 'reward', 'done', 'info')}
    for t in range(T):
        rows = {k: [] for k in out}
        for e, env in enumerate(envs):
            a = actions[t, e]
            act = [rh.onehot(x) for x in a] if a.ndim == 1 else [np.array(x, dtype=np.float64) for x in a]
            obs, ids, node, adj, rew, done, info = env.step(act)
            assert all(np.array_equal(adj[0], x) for x in adj)
            assert [int(i[0]) for i in ids] == list(range(len(ids)))
            rows['obs'].append(np.array(obs)); rows['node_obs'].append(np.array(node))
            rows['adj'].append(np.array(adj[0])); rows['reward'].append(np.array(rew, dtype=np.float64))
            rows['done'].append(np.array(done)); rows['info'].append(rh.info_array(info, info_keys))
        for k in out:
            out[k].append(np.stack(rows[k]))
    res = {k: np.stack(v) for k, v in out.items()}
    res.update({'init_' + k: v for k, v in stack_states(init, fields).items()})
    res.update({'final_' + k: v for k, v in stack_states([rh.capture_state(e) for e in envs], fields).items()})
    res['actions'] = actions
    res['args'] = np.array(json.dumps(vars(args)))
    res['info_keys'] = np.array(info_keys)
    return res


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024))


def gen_traj():
    rs = np.random.RandomState(2024)
    cases = [  # name, N, O, W, n_envs, T
        ('n3', 3, 3, 0, 4, 25), ('n3w2', 3, 3, 2, 4, 25), ('n1', 1, 0, 0, 2, 10),
        ('n2o1w1', 2, 1, 1, 2, 12), ('n10', 10, 3, 0, 2, 25), ('n32', 32, 8, 0, 1, 6)]
    for name, N, O, W, n, T in cases:
        args = rh.make_args(num_agents=N, num_landmarks=N, num_obstacles=O, num_walls=W)
        actions = rs.randint(0, 5, size=(T, n, N)).astype(np.int64)
        save('traj_%s.npz' % name, run_traj(args, [11 + 7 * e for e in range(n)], actions))
    # long horizon past episode_length (done stays True; the bare env never auto-resets)
    args = rh.make_args(num_agents=3, num_landmarks=3, num_obstacles=3, episode_length=5)
    actions = rs.randint(0, 5, size=(9, 2, 3)).astype(np.int64)
    save('traj_n3_ep5.npz', run_traj(args, [5, 6], actions))
    # general (non one-hot) float actions: environment.py:303-304 adds a[1]-a[2], a[3]-a[4]
    args = rh.make_args(num_agents=3, num_landmarks=3, num_obstacles=2)
    actions = rs.uniform(0, 1, size=(8, 2, 3, 5)).astype(np.float32).astype(np.float64)  # float32-representable
    save('traj_n3_float.npz', run_traj(args, [21, 22], actions))
    # non-default reward knobs
    args = rh.make_args(num_agents=4, num_landmarks=4, num_obstacles=2, num_walls=1, goal_rew=7.5,
                        collision_rew=3, fair_rew=2.5, zeroshift=1.5, min_dist_thresh=0.1, max_speed=1.2)
    actions = rs.randint(0, 5, size=(15, 2, 4)).astype(np.int64)
    save('traj_n4_knobs.npz', run_traj(args, [31, 32], actions))
    # crafted: agents on / near their goals (arrival, frozen Dg), overlapping agents and an
    # agent inside an obstacle (collision penalties, stiff contact), agent hugging a wall end-cap.
    args = rh.make_args(num_agents=3, num_landmarks=3, num_obstacles=2, num_walls=1)
    c0 = dict(agent_pos=np.array([[0.30, 0.30], [0.36, 0.30], [-0.5, 0.2]]),
              agent_vel=np.zeros((3, 2)), landmark_pos=np.array([[0.32, 0.31], [0.0, 0.0], [-0.5, 0.23]]),
              obstacle_pos=np.array([[-0.45, 0.22], [0.7, -0.7]]), goal_match=np.array([0, 1, 2]),
              wall_axis=np.array([0.6]), wall_e0=np.array([-0.3]), wall_e1=np.array([0.3]),
              wall_orient=np.array([0]))
    c1 = dict(agent_pos=np.array([[0.33, 0.58], [-0.2, 0.63], [0.1, -0.9]]),
              agent_vel=np.array([[0.5, 0.3], [0.0, -0.4], [1.9, 0.9]]),
              landmark_pos=np.array([[0.1, -0.86], [-0.2, 0.2], [0.4, 0.4]]),
              obstacle_pos=np.array([[0.12, -0.93], [0.5, 0.5]]), goal_match=np.array([2, 1, 0]),
              wall_axis=np.array([0.6]), wall_e0=np.array([-0.3]), wall_e1=np.array([0.3]),
              wall_orient=np.array([0]))
    actions = rs.randint(0, 5, size=(15, 2, 3)).astype(np.int64)
    actions[:3, 0, 0] = 0
    save('traj_crafted.npz', run_traj(args, [c0, c1], actions, crafted=True))


def gen_global():
    """--graph_feat_type global (navigation_graph.py:981-1009): 7-wide absolute node features."""
    rs = np.random.RandomState(77)
    for name, N, O, n, T in (('n3', 3, 3, 3, 12), ('n10', 10, 2, 2, 8)):
        args = rh.make_args(num_agents=N, num_landmarks=N, num_obstacles=O, graph_feat_type='global')
        actions = rs.randint(0, 5, size=(T, n, N)).astype(np.int64)
        save('traj_%s_global.npz' % name, run_traj(args, [41 + 3 * e for e in range(n)], actions))


def gen_formation():
    """fair_graph_formation (BASELINE config 4 shapes: N=10, L=1, O=3, W=2 -> E=16) + edge cases."""
    rs = np.random.RandomState(77)
    kw = dict(info_keys=fo.INFO_KEYS, fields=fo.State.FIELDS)
    cases = [  # name, N, L, O, thr, n_envs, T
        ('n3', 3, 1, 3, 0.05, 3, 25), ('n10', 10, 1, 3, 0.05, 2, 25), ('n5l3', 5, 3, 2, 0.05, 2, 12),
        ('n4_thr04', 4, 1, 2, 0.4, 3, 20), ('n3_thr07', 3, 1, 1, 0.7, 3, 20), ('n1', 1, 1, 0, 0.05, 2, 8)]
    for name, N, L, O, thr, n, T in cases:
        args = rh.make_args(scenario_name='fair_graph_formation', num_agents=N, num_landmarks=L, num_obstacles=O,
                            min_dist_thresh=thr)
        actions = rs.randint(0, 5, size=(T, n, N)).astype(np.int64)
        save('form_%s.npz' % name, run_traj(args, [41 + 3 * e for e in range(n)], actions, **kw))
    # agents parked on the circle (arrival / formation_complete / occupied slots) and near the walls
    args = rh.make_args(scenario_name='fair_graph_formation', num_agents=4, num_landmarks=1, num_obstacles=1)
    ang = np.array([0.3, 0.3 + np.pi / 2, 0.3 + np.pi, 4.0])
    ring = 0.5 * np.stack([np.cos(ang), np.sin(ang)], axis=1)
    c0 = dict(agent_pos=ring + np.array([0.1, -0.05]), agent_vel=np.zeros((4, 2)), landmark_pos=np.array([[0.1, -0.05]]),
              obstacle_pos=np.array([[0.9, 0.9]]))
    c1 = dict(agent_pos=np.array([[0.52, 0.1], [-0.47, -0.2], [0.0, 0.49], [0.3, 0.3]]),
              agent_vel=np.array([[0.4, 0.0], [0.0, 0.3], [-0.2, 0.1], [0.0, 0.0]]), landmark_pos=np.array([[0.0, 0.0]]),
              obstacle_pos=np.array([[0.33, 0.31]]), wall_axis=np.array([0.5, -0.5]), wall_e0=np.array([-0.3, -0.3]),
              wall_e1=np.array([0.3, 0.3]))
    actions = rs.randint(0, 5, size=(12, 2, 4)).astype(np.int64)
    actions[:4, 0] = 0
    save('form_crafted.npz', run_traj(args, [c0, c1], actions, crafted=True, **kw))
    # the reference's GraphDummyVecEnv x 4 incl. auto-resets on the NumPy global stream
    from onpolicy.envs.env_wrappers import GraphDummyVecEnv
    from multiagent.MPE_env import GraphMPEEnv
    n, N, seed, T = 4, 3, 2, 55
    args = rh.make_args(scenario_name='fair_graph_formation', num_agents=N, num_landmarks=1, num_obstacles=2)

    def fn(r):
        def init():
            env = GraphMPEEnv(args)
            env.seed(seed + r * 1000)
            return env
        return init
    acts = np.eye(5)[rs.randint(0, 5, size=(T, n, N))]
    np.random.seed(seed)
    venv = GraphDummyVecEnv([fn(r) for r in range(n)])
    r0 = venv.reset()
    d = dict(reset_obs=r0[0], reset_id=r0[1], reset_node_obs=r0[2], reset_adj=r0[3][:, 0], actions=acts,
             args=np.array(json.dumps(vars(args))), seed=np.int64(seed), info_keys=np.array(fo.INFO_KEYS))
    keys = ('obs', 'agent_id', 'node_obs', 'adj', 'reward', 'done')
    rec = {k: [] for k in keys + ('info', 'reset_count')}
    for t in range(T):
        res = venv.step(acts[t])
        for k, v in zip(keys, res[:6]):
            rec[k].append(v[:, 0] if k == 'adj' else v)
        rec['info'].append(np.stack([rh.info_array(res[6][e], fo.INFO_KEYS) for e in range(n)]))
        rec['reset_count'].append(res[7])
    d.update({k: np.stack(v) for k, v in rec.items()})
    save('form_dummy4.npz', d)


def gen_fairnav():
    """nav_fairassign_fairrew_formation_graph (the scenario of the shipped FA / FA+FR weights)."""
    SC = 'nav_fairassign_fairrew_formation_graph'
    rs = np.random.RandomState(311)
    kw = dict(info_keys=fnv.INFO_KEYS, fields=fnv.State.FIELDS)
    cases = [  # name, N, O, W, thr, min_obs_dist, n_envs, T
        ('n3', 3, 3, 0, 0.05, 0.5, 3, 25), ('n10', 10, 3, 0, 0.05, 0.5, 2, 20), ('n4w2', 4, 2, 2, 0.25, 0.6, 3, 30),
        ('n7_thr035', 7, 1, 0, 0.35, 0.4, 2, 30), ('n3_thr04', 3, 2, 0, 0.4, 0.9, 3, 30), ('n2', 2, 0, 1, 0.5, 1.5, 2, 10)]
    for name, N, O, W, thr, mod, n, T in cases:
        args = rh.make_args(scenario_name=SC, num_agents=N, num_landmarks=N, num_obstacles=O, num_walls=W,
                            min_dist_thresh=thr, min_obs_dist=mod)
        actions = rs.randint(0, 5, size=(T, n, N)).astype(np.int64)
        save('fnav_%s.npz' % name, run_traj(args, [61 + 5 * e for e in range(n)], actions, **kw))
    from onpolicy.envs.env_wrappers import GraphDummyVecEnv
    from multiagent.MPE_env import GraphMPEEnv
    n, N, seed, T = 3, 3, 1, 70
    args = rh.make_args(scenario_name=SC, num_agents=N, num_landmarks=N, num_obstacles=2, min_dist_thresh=0.3)

    def fn(r):
        def init():
            env = GraphMPEEnv(args)
            env.seed(seed + r * 1000)
            return env
        return init
    acts = np.eye(5)[rs.randint(0, 5, size=(T, n, N))]
    np.random.seed(seed)
    venv = GraphDummyVecEnv([fn(r) for r in range(n)])
    r0 = venv.reset()
    d = dict(reset_obs=r0[0], reset_id=r0[1], reset_node_obs=r0[2], reset_adj=r0[3][:, 0], actions=acts,
             args=np.array(json.dumps(vars(args))), seed=np.int64(seed), info_keys=np.array(fnv.INFO_KEYS))
    keys = ('obs', 'agent_id', 'node_obs', 'adj', 'reward', 'done')
    rec = {k: [] for k in keys + ('info', 'reset_count')}
    for t in range(T):
        res = venv.step(acts[t])
        for k, v in zip(keys, res[:6]):
            rec[k].append(v[:, 0] if k == 'adj' else v)
        rec['info'].append(np.stack([rh.info_array(res[6][e], fnv.INFO_KEYS) for e in range(n)]))
        rec['reset_count'].append(res[7])
    d.update({k: np.stack(v) for k, v in rec.items()})
    save('fnav_dummy3.npz', d)


def gen_cfg1():
    from onpolicy.envs.env_wrappers import GraphDummyVecEnv
    from multiagent.MPE_env import GraphMPEEnv
    n, N, seed, T = 8, 3, 1, 60
    args = rh.make_args(num_agents=N, num_landmarks=N, num_obstacles=3)

    def fn(r):
        def init():
            env = GraphMPEEnv(args)
            env.seed(seed + r * 1000)  # onpolicy/scripts/train_mpe.py:31
            return env
        return init
    rs = np.random.RandomState(99)
    acts = np.eye(5)[rs.randint(0, 5, size=(T, n, N))]
    np.random.seed(seed)
    venv = GraphDummyVecEnv([fn(r) for r in range(n)])
    r0 = venv.reset()
    d = dict(reset_obs=r0[0], reset_id=r0[1], reset_node_obs=r0[2], reset_adj=r0[3][:, 0],
             actions=acts, args=np.array(json.dumps(vars(args))), seed=np.int64(seed),
             info_keys=np.array(INFO_KEYS))
    keys = ('obs', 'agent_id', 'node_obs', 'adj', 'reward', 'done')
    rec = {k: [] for k in keys + ('info', 'reset_count')}
    for t in range(T):
        res = venv.step(acts[t])
        assert len(res) == 8 and res[6].dtype == object
        for k, v in zip(keys, res[:6]):
            rec[k].append(v[:, 0] if k == 'adj' else v)
        assert np.array_equal(res[3], np.broadcast_to(res[3][:, :1], res[3].shape))
        rec['info'].append(np.stack([rh.info_array(res[6][e], INFO_KEYS) for e in range(n)]))
        rec['reset_count'].append(res[7])
    d.update({k: np.stack(v) for k, v in rec.items()})
    save('cfg1_dummy8.npz', d)


def gen_kat_world():
    """Single World.step() answers on hand-built worlds."""
    from multiagent.core import World, Agent, Landmark, Wall

    def world(agents, obstacles=(), walls=()):
        w = World(); w.cache_dists = True; w.dim_c = 2
        for p, v in agents:
            a = Agent(); a.state.p_pos = np.array(p, float); a.state.p_vel = np.array(v, float)
            a.silent = True; a.max_speed = 2; a.action.u = np.zeros(2); a.name = 'agent %d' % len(w.agents)
            w.agents.append(a)
        for p in obstacles:
            o = Landmark(); o.state.p_pos = np.array(p, float); o.state.p_vel = np.zeros(2)
            o.collide = True; o.movable = False; w.obstacles.append(o)
        for orient, axis, ends in walls:
            x = Wall(orient=orient, axis_pos=axis, endpoints=ends, width=0.1)
            x.collide = True; x.movable = False
            x.state.p_pos = np.array([0.0, axis]) if orient == 'H' else np.array([axis, 0.0])
            x.state.p_vel = np.zeros(2); w.walls.append(x)
        w.calculate_distances()
        return w
    kats = []

    def rec(w, us, steps=1):
        rows = []
        for _ in range(steps):
            for a, u in zip(w.agents, us):
                a.action.u = np.array(u, float)
            w.step()
            rows.append(np.array([[*a.state.p_pos, *a.state.p_vel, a.state.p_dist] for a in w.agents]))
        return np.stack(rows)
    specs = [
        dict(agents=[((0, 0), (0, 0))], us=[(5, 0)], steps=2),
        dict(agents=[((0, 0), (0, 0)), ((0.08, 0), (0, 0))], us=[(0, 0), (0, 0)]),
        dict(agents=[((0, 0), (0.3, -0.2))], obstacles=[(0.09, 0.06)], us=[(0, 0)]),
        dict(agents=[((0.1, 0.42), (0, 0)), ((0.43, 0.45), (0, 0))], walls=[('H', 0.5, (-0.4, 0.4))], us=[(0, 0), (0, 0)]),
        dict(agents=[((0, 0), (3, 4))], us=[(5, 5)]),
        dict(agents=[((-0.52, 0.43), (0, 0)), ((0.2, -0.44), (0.1, 0))], walls=[('V', -0.5, (-0.4, 0.4))], us=[(0, 5), (-5, 0)], steps=3),
    ]
    d = {}
    for i, sp in enumerate(specs):
        w = world(sp['agents'], sp.get('obstacles', ()), sp.get('walls', ()))
        d['kat%d_agents' % i] = np.array([[*p, *v] for p, v in sp['agents']], float)
        d['kat%d_obstacles' % i] = np.array(sp.get('obstacles', ()), float).reshape(-1, 2)
        d['kat%d_walls' % i] = np.array([[0 if o == 'H' else 1, ax, e[0], e[1]] for o, ax, e in sp.get('walls', ())], float).reshape(-1, 4)
        d['kat%d_u' % i] = np.array(sp['us'], float)
        d['kat%d_out' % i] = rec(w, sp['us'], sp.get('steps', 1))
    d['n_kats'] = np.int64(len(specs))
    save('kat_world.npz', d)


if __name__ == '__main__':
    assert rh.available(), 'needs /root/reference (build container only)'
    which = sys.argv[1:] or ['kat', 'cfg1', 'traj', 'global', 'formation', 'fairnav']
    if 'kat' in which:
        gen_kat_world()
    if 'cfg1' in which:
        gen_cfg1()
    if 'traj' in which:
        gen_traj()
    if 'global' in which:
        gen_global()
    if 'formation' in which:
        gen_formation()
    if 'fairnav' in which:
        gen_fairnav()
