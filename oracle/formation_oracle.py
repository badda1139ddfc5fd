"""TEST INFRASTRUCTURE ONLY — NumPy float64 restatement (oracle) of the ``fair_graph_formation``
scenario (BASELINE config 4).  Never imported by the product.

Physics, distance matrix and the fairness-scalar rule are shared with ``nav_oracle`` (the reference
scenarios share ``multiagent/core.py`` and copy the fairness code); what differs is restated here
from ``multiagent/custom_scenarios/fair_graph_formation.py`` (cited per function as ``ff:<line>``).

The per-agent pass of one env step is stateful in the reference (``environment.py:832-864`` calls
observation / reward / graph_observation / info for agent 0, then agent 1, ... and the scenario
mutates ``expected_poses`` / ``expected_poses_occupied`` inside those calls), so it is emulated
sequentially per env in exactly that order — this is the oracle, clarity over speed.
Parity status: pinned by ``tests/golden/form_*.npz`` (outputs of the reference itself).
SciPy's ``linear_sum_assignment`` is the same third-party routine the reference calls (ff:16).
"""
from dataclasses import dataclass, fields

import numpy as np
from scipy.optimize import linear_sum_assignment

from . import nav_oracle as no
from .nav_oracle import DT, ENTITY_SIZE, WALL_WIDTH

TARGET_RADIUS = 0.5  # ff:105
INFO_KEYS = ('Dist_to_goal', 'Time_req_to_goal', 'Num_agent_collisions', 'Num_obst_collisions',
             'Distance_mean', 'Distance_variance', 'Mean_by_variance', 'Dists_traveled', 'Time_taken',
             'Formation_dist', 'Min_time_to_goal', 'individual_reward')  # ff:484-499, environment.py:861


@dataclass
class Config(no.Config):
    scenario_name: str = 'fair_graph_formation'
    num_landmarks: int = 1
    num_walls: int = 2  # hard-coded in the scenario, ff:184 (args.num_walls is ignored)

    @classmethod
    def from_args(cls, args):
        kw = {f.name: getattr(args, f.name) for f in fields(cls) if hasattr(args, f.name)}
        kw['num_walls'] = 2
        return cls(**kw)

    @property
    def obs_dim(self): return 6
    @property
    def node_feat(self): return 12


class State(no.State):
    FIELDS = no.State.FIELDS + ('slot_pos', 'slot_occ', 'slot_delta', 'formation_done')

    def __init__(self, cfg, n):
        super().__init__(cfg, n)
        N = cfg.N
        self.slot_pos = np.zeros((n, N, 2))     # scenario.expected_poses
        self.slot_occ = np.zeros((n, N))        # scenario.expected_poses_occupied
        self.slot_delta = np.zeros((n, N))      # scenario.delta_dists
        self.formation_done = np.zeros((n, N))  # world.formation_complete


def find_angle(p):
    """ff:35-40"""
    a = np.arctan2(p[1], p[0])
    return a + 2 * np.pi if a < 0 else a


def expected_poses(cfg, agent_pos, landmark0):
    """ff:394-413 / ff:630-648: N slots on a circle of radius 0.5 about landmark 0, anchored at the
    smallest polar angle of the agents."""
    N = cfg.N
    theta_min = min(find_angle(p - landmark0) for p in agent_pos)
    sep = (2 * np.pi) / N
    return np.array([landmark0 + TARGET_RADIUS * np.array([np.cos(theta_min + i * sep), np.sin(theta_min + i * sep)])
                     for i in range(N)])


def pair_dists(agent_pos, slots):
    """ff:650-655: dists[a, k] = |x_a - P_k| (np.linalg.norm)."""
    d = agent_pos[:, None, :] - slots[None, :, :]
    return np.sqrt(np.sum(np.square(d), axis=-1))


def obstacle_hit(cfg, st, e, pos):
    """ff:503-531 is_obstacle_collision: obstacles at 1.05 (s+s); wall boxes WITHOUT the 1.05 factors."""
    s = ENTITY_SIZE
    for o in range(cfg.O):
        if np.linalg.norm(st.obstacle_pos[e, o] - pos) < 1.05 * (s + s):
            return True
    for w in range(cfg.W):
        axis, e0, e1 = st.wall_axis[e, w], st.wall_e0[e, w], st.wall_e1[e, w]
        pperp, ppar = (pos[1], pos[0]) if st.wall_orient[e, w] == 0 else (pos[0], pos[1])
        if axis - s / 2 <= pperp <= axis + s / 2 and e0 - s / 2 <= ppar <= e1 + s / 2:
            return True
    return False


class _EnvPass:
    """Sequential agent loop of one env (environment.py:832-864 / :892-897) on fixed positions."""

    def __init__(self, cfg, st, e):
        self.cfg, self.st, self.e = cfg, st, e
        self.x = st.agent_pos[e]
        self.v = st.agent_vel[e]
        self._hung = {}

    def hungarian(self):
        """linear_sum_assignment(dists)[1] for the CURRENT slots (positions are fixed in the pass)."""
        key = self.st.slot_pos[self.e].tobytes()
        if key not in self._hung:
            self._hung[key] = linear_sum_assignment(pair_dists(self.x, self.st.slot_pos[self.e]))[1]
        return self._hung[key]

    def goal_of(self, pos, ego):
        """The three-way branch of ff:707-739 (observation) == ff:916-943 (agent rows of the graph).
        Returns (goal, flag) and mutates the occupancy vector like the reference."""
        st, e, cfg = self.st, self.e, self.cfg
        P, occ = st.slot_pos[e], st.slot_occ[e]
        d = np.array([np.linalg.norm(pos - l) for l in P])
        k = int(np.argmin(d))
        if d[k] < cfg.min_dist_thresh:
            occ[k] = 1
            return P[k].copy(), occ[k]
        if (occ == 0).any():
            g = self.hungarian()
            return P[g[ego]].copy(), occ[g[ego]]
        occ[:] = 0
        return pos.copy(), occ[ego]

    def observation(self, i):
        goal, flag = self.goal_of(self.x[i], i)
        # ff:740-741: list + ndarray broadcasts -> concat(v, x, goal - x) + flag
        return np.concatenate([self.v[i], self.x[i], goal - self.x[i]]) + flag

    def graph_observation(self, i):
        """ff:810-850 + ff:896-971 -> (E, 12)."""
        cfg, st, e = self.cfg, self.st, self.e
        N, L, O, W = cfg.N, cfg.L, cfg.O, cfg.W
        rows = []
        xi, vi = self.x[i], self.v[i]
        for a in range(N):
            goal, flag = self.goal_of(self.x[a], i)  # NB: branch (b) indexes with the EGO id (ff:936-937)
            rp = self.x[a] - xi
            rows.append(np.hstack([self.v[a] - vi, rp, goal - xi, [flag], rp, rp, 0]))
        for l in range(L):
            rp = st.landmark_pos[e, l] - xi
            rows.append(np.hstack([-vi, rp, rp, [1], rp, rp, 1]))
        for o in range(O):
            rp = st.obstacle_pos[e, o] - xi
            rows.append(np.hstack([-vi, rp, rp, [1], rp, rp, 2]))
        wp = st.wall_pos()[e]
        for w in range(W):
            rp = wp[w] - xi
            oc = np.array([st.wall_e0[e, w], st.wall_axis[e, w] + WALL_WIDTH / 2]) - xi
            dc = np.array([st.wall_e1[e, w], st.wall_axis[e, w] - WALL_WIDTH / 2]) - xi
            rows.append(np.hstack([-vi, rp, rp, [1], oc, dc, 3]))
        return np.array(rows)


def env_step(cfg, st, actions):
    """MultiAgentGraphEnv.step (environment.py:816-877) with the formation scenario callbacks."""
    n, N = st.agent_pos.shape[:2]
    st.cur_step = st.cur_step + 1
    no.world_step(cfg, st, no.decode_actions(cfg, actions))
    dist = no.distance_matrix(st)
    obs = np.zeros((n, N, 6)); node = np.zeros((n, N, cfg.E, 12)); rew = np.zeros((n, N))
    info = np.zeros((n, N, len(INFO_KEYS)))
    thr = cfg.min_dist_thresh
    for e in range(n):
        ps = _EnvPass(cfg, st, e)
        x = st.agent_pos[e]
        Dg, Tr = st.dists_to_goal[e], st.times_required[e]
        d_mean, d_std = np.mean(Dg), np.std(Dg)  # statistics left by the previous info call (ff:477-478)
        for i in range(N):
            obs[e, i] = ps.observation(i)
            # ---- reward ff:622-700
            if Dg[i] == -1:
                f = np.mean(st.p_dist[e]) / (np.std(st.p_dist[e]) + 0.0001)
            else:
                f = d_mean / (d_std + 0.0001)
            if i == 0:
                st.slot_pos[e] = expected_poses(cfg, x, st.landmark_pos[e, 0])
                dm = pair_dists(x, st.slot_pos[e])
                st.slot_occ[e] = np.any(dm < thr, axis=0).astype(int)
                ri, ci = linear_sum_assignment(dm)
                st.slot_delta[e] = dm[ri, ci]
            r = cfg.goal_rew if st.slot_delta[e, i] < thr else -st.slot_delta[e, i]
            ag_hits = sum(1 for a in range(N) if a != i and np.linalg.norm(x[a] - x[i]) < 1.05 * 2 * ENTITY_SIZE)
            ob_hit = obstacle_hit(cfg, st, e, x[i])
            r = r - cfg.collision_rew * ag_hits - (cfg.collision_rew if ob_hit else 0)
            r = r + cfg.fair_rew * np.tanh(f - 5.0)  # hard-coded 5.0, no floor (ff:693-696)
            r = float(np.clip(r, -2 * cfg.collision_rew, cfg.goal_rew + cfg.fair_rew))
            rew[e, i] = r
            node[e, i] = ps.graph_observation(i)
            # ---- info ff:441-501
            dists = np.array([np.linalg.norm(x[i] - l) for l in st.slot_pos[e]])
            fd = np.linalg.norm(x[i] - st.landmark_pos[e, 0])
            if 0.95 * TARGET_RADIUS < fd < 1.05 * TARGET_RADIUS:
                st.formation_done[e, i] = 1
                if Tr[i] == -1:
                    Tr[i] = st.cur_step[e] * DT
                    Dg[i] = st.p_dist[e, i]
            st.dist_left[e, i] = np.min(dists)
            if Tr[i] == -1:
                Dg[i] = st.p_dist[e, i]
            if ob_hit:
                st.num_obst_coll[e, i] += 1
            st.num_agent_coll[e, i] += ag_hits
            d_mean, d_std = np.mean(Dg), np.std(Dg)
            info[e, i] = [st.dist_left[e, i], Tr[i], st.num_agent_coll[e, i], st.num_obst_coll[e, i], d_mean, d_std,
                          d_mean / (d_std + 0.0001), Dg[i], 0.0, st.formation_done[e, i], st.min_time[e, i], r]
    done = np.broadcast_to((st.cur_step >= cfg.episode_length)[:, None], (n, N)).copy()
    return dict(obs=obs, node_obs=node, adj=dist, reward=rew, done=done, info=info)


def observe_reset(cfg, st, envs=None):
    """environment.py:882-898: obs + graph obs of every agent in order (mutates the occupancy flags)."""
    n, N = st.agent_pos.shape[:2]
    obs = np.zeros((n, N, 6)); node = np.zeros((n, N, cfg.E, 12))
    for e in (range(n) if envs is None else envs):
        ps = _EnvPass(cfg, st, e)
        for i in range(N):
            obs[e, i] = ps.observation(i)
            node[e, i] = ps.graph_observation(i)
    return dict(obs=obs, node_obs=node, adj=no.distance_matrix(st))


def reset_env(cfg, st, e, rng):
    """ff:212-248 reset_world + ff:251-439 random_scenario for env ``e``."""
    N, L, O, W = cfg.N, cfg.L, cfg.O, cfg.W
    ws = cfg.world_size
    st.cur_step[e] = 0
    st.times_required[e] = -1; st.dists_to_goal[e] = -1; st.dist_left[e] = -1
    st.num_obst_coll[e] = 0; st.num_agent_coll[e] = 0
    st.formation_done[e] = 0
    st.p_dist[e] = 0; st.time[e] = 0
    for o in range(O):
        st.obstacle_pos[e, o] = 0.8 * rng.uniform_pair(-ws / 2, ws / 2)
    wall_position = rng.uniform(0.2, 0.9)
    wall_axis = [wall_position * ws / 2, -wall_position * ws / 2]
    for w in range(W):  # always vertical, no orientation draw (ff:276)
        st.wall_orient[e, w] = 1
        st.wall_e0[e, w] = -st.wall_length[e]; st.wall_e1[e, w] = st.wall_length[e]
        st.wall_axis[e, w] = wall_axis[w]
    thr = 1.05 * 2 * ENTITY_SIZE
    k = tries = 0
    while k < N:
        p = rng.uniform_pair(-ws / 2, ws / 2)
        tries += 1
        bad = obstacle_hit(cfg, st, e, p)
        if not bad and k:
            bad = bool((np.sqrt(np.sum(np.square(st.agent_pos[e, :k] - p), axis=-1)) < thr).any())
        if not bad or tries >= no.MAX_TRIES:
            st.agent_pos[e, k] = p; st.agent_vel[e, k] = 0
            k += 1; tries = 0
    k = tries = 0
    while k < L:
        p = 0.5 * rng.uniform_pair(-ws / 2, ws / 2)  # ff:363
        tries += 1
        bad = obstacle_hit(cfg, st, e, p)
        if not bad and k:
            bad = bool((np.sqrt(np.sum(np.square(st.landmark_pos[e, :k] - p), axis=-1)) < thr).any())
        if not bad or tries >= no.MAX_TRIES:
            st.landmark_pos[e, k] = p
            k += 1; tries = 0
    st.slot_pos[e] = expected_poses(cfg, st.agent_pos[e], st.landmark_pos[e, 0])
    st.slot_occ[e] = 0
    if cfg.max_speed is not None:  # ff:573-580: distance to the agent's OWN slot index
        st.min_time[e] = np.sqrt(np.sum(np.square(st.agent_pos[e] - st.slot_pos[e]), axis=-1)) / cfg.max_speed


class OracleFormationVecEnv:
    """GraphMPEEnv x n + auto-reset wrapper semantics for the formation scenario (see nav_oracle)."""

    def __init__(self, cfg, n, seeds=None, mode='dummy', streams=None):
        self.cfg, self.n, self.mode, self.streams = cfg, n, mode, streams
        self.st = State(cfg, n)
        self.episode = np.zeros(n, dtype=np.int64)
        for e in range(n):
            rng = self._rng(e)
            self.st.wall_length[e] = no.draw_wall_length(cfg, rng)  # ff:180-182
            reset_env(cfg, self.st, e, rng)
            self.episode[e] += 1
            if streams is None and seeds is not None:
                np.random.seed(int(seeds[e]))
        # MultiAgentBaseEnv.__init__ / set_graph_obs_space call observation + graph_observation for
        # every agent once more (environment.py:112-117, :788-790): they mutate the occupancy flags.
        for e in range(n):
            ps = _EnvPass(cfg, self.st, e)
            for i in range(cfg.N):
                ps.observation(i)
            for i in range(cfg.N):
                ps.graph_observation(i)

    def _rng(self, e):
        if self.streams is not None:
            return self.streams(e, int(self.episode[e]))
        return no.NumpyGlobalStream()

    def _reset_one(self, e):
        reset_env(self.cfg, self.st, e, self._rng(e))
        self.episode[e] += 1

    def _agent_id(self):
        return np.tile(np.arange(self.cfg.N, dtype=np.int64)[None, :, None], (self.n, 1, 1))

    def reset(self):
        for e in range(self.n):
            self._reset_one(e)
        o = observe_reset(self.cfg, self.st)
        adj = np.broadcast_to(o['adj'][:, None], (self.n, self.cfg.N) + o['adj'].shape[1:]).copy()
        return o['obs'], self._agent_id(), o['node_obs'], adj

    def step(self, actions):
        out = env_step(self.cfg, self.st, actions)
        done_all = out['done'].all(axis=1)
        reset_count = 0
        if done_all.any():
            idx = np.nonzero(done_all)[0]
            for e in idx:
                reset_count = 1
                self._reset_one(e)
            o = observe_reset(self.cfg, self.st, idx)
            for k in ('obs', 'node_obs'):
                out[k][idx] = o[k][idx]
            out['adj'][idx] = o['adj'][idx]
        N = self.cfg.N
        adj = np.broadcast_to(out['adj'][:, None], (self.n, N) + out['adj'].shape[1:]).copy()
        res = (out['obs'], self._agent_id(), out['node_obs'], adj, out['reward'], out['done'], out['info'])
        return res + (reset_count,) if self.mode == 'dummy' else res
