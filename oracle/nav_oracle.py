"""TEST INFRASTRUCTURE ONLY — NumPy float64 restatement (oracle) of the Fair-MARL rollout
hot path for the ``navigation_graph`` scenario, batched over environments.

Nothing here is shipped or measured as the product: ``fair_marl_amd`` never imports it.
It exists to (i) check the HIP path in ``tests/``, (ii) be checked itself against golden
vectors captured from the imported reference (``tests/golden/gen_golden.py``), and (iii)
serve as the ``cpu_baseline`` leg of ``bench.py``.

Every function cites the reference lines (relative to ``/root/reference``) it restates.
Parity status: pinned — ``tests/test_oracle_golden.py`` replays the committed fixtures
(outputs of the reference itself run in the build container) through this file.
The fair-assignment solver is the exception (Gurobi absent): see ``oracle/lexifair.py``.
"""
from dataclasses import dataclass, field, fields

import numpy as np

from .lexifair import lexifair

# world constants -- multiagent/core.py:153-161, :68, :38
DT = 0.1
DAMPING = 0.25
CONTACT_FORCE = 3e2
CONTACT_MARGIN = 2e-2
WALL_CONTACT_FORCE = 2.2e2
WALL_CONTACT_MARGIN = 2.4e-2
ENTITY_SIZE = 0.05
WALL_WIDTH = 0.1
SENSITIVITY = 5.0  # multiagent/environment.py:307

INFO_KEYS = ('Dist_to_goal', 'Time_req_to_goal', 'Num_agent_collisions', 'Num_obst_collisions',
             'Distance_mean', 'Distance_variance', 'Mean_by_variance', 'Dists_traveled',
             'Time_taken', 'Time_mean', 'Time_stddev', 'Time_mean_by_stddev',
             'Min_time_to_goal', 'individual_reward')  # navigation_graph.py:625-647, environment.py:861


@dataclass
class Config:
    """Scenario arguments (defaults: onpolicy/config.py, scripts/train_mpe.py:71-106)."""
    scenario_name: str = 'navigation_graph'
    num_agents: int = 3
    num_landmarks: int = 3
    num_obstacles: int = 3
    num_walls: int = 0
    world_size: float = 2
    max_speed: float = 2
    collision_rew: float = 5
    goal_rew: float = 5
    min_dist_thresh: float = 0.05
    fair_wt: float = 1
    fair_rew: float = 1
    zeroshift: float = 5
    max_edge_dist: float = 1
    episode_length: int = 25
    collaborative: bool = False
    use_dones: bool = False
    graph_feat_type: str = 'relative'
    num_scripted_agents: int = 0

    @classmethod
    def from_args(cls, args):
        kw = {f.name: getattr(args, f.name) for f in fields(cls) if hasattr(args, f.name)}
        return cls(**kw)

    @property
    def N(self): return self.num_agents
    @property
    def L(self): return self.num_landmarks
    @property
    def O(self): return self.num_obstacles
    @property
    def W(self): return self.num_walls
    @property
    def E(self): return self.N + self.L + self.O + self.W
    @property
    def obs_dim(self): return 7
    @property
    def node_feat(self): return 7 if self.graph_feat_type == 'global' else 11


class State:
    """Batched SoA state of ``n`` independent worlds (reference: one ``World`` object each)."""

    def __init__(self, cfg, n):
        N, L, O, W = cfg.N, cfg.L, cfg.O, cfg.W
        z = np.zeros
        self.agent_pos = z((n, N, 2)); self.agent_vel = z((n, N, 2))
        self.p_dist = z((n, N)); self.time = z((n, N))
        self.landmark_pos = z((n, L, 2)); self.obstacle_pos = z((n, O, 2))
        self.wall_axis = z((n, W)); self.wall_e0 = z((n, W)); self.wall_e1 = z((n, W))
        self.wall_orient = z((n, W), dtype=np.int64)  # 0 = 'H', 1 = 'V'
        self.wall_length = z((n,))
        self.goal_match = np.tile(np.arange(N), (n, 1))  # navigation_graph.py:110
        self.dists_to_goal = -np.ones((n, N)); self.times_required = -np.ones((n, N))
        self.dist_left = -np.ones((n, N))
        self.num_obst_coll = z((n, N)); self.num_agent_coll = z((n, N))
        self.min_time = np.full((n, N), np.inf)  # core.py:125
        self.cur_step = z((n,), dtype=np.int64)

    FIELDS = ('agent_pos', 'agent_vel', 'p_dist', 'time', 'landmark_pos', 'obstacle_pos',
              'wall_axis', 'wall_e0', 'wall_e1', 'wall_orient', 'wall_length', 'goal_match',
              'dists_to_goal', 'times_required', 'dist_left', 'num_obst_coll', 'num_agent_coll',
              'min_time', 'cur_step')

    def copy(self):
        c = object.__new__(type(self))
        for k, v in self.__dict__.items():
            setattr(c, k, v.copy())
        return c

    def wall_pos(self):
        """Wall 'sphere' centre: (0, axis) for 'H', (axis, 0) for 'V' (navigation_graph.py:309-324)."""
        zero = np.zeros_like(self.wall_axis)
        x = np.where(self.wall_orient == 0, zero, self.wall_axis)
        y = np.where(self.wall_orient == 0, self.wall_axis, zero)
        return np.stack([x, y], axis=-1)

    def entity_pos(self):
        """core.py:179-186 order: agents, landmarks, obstacles, (wall_obstacles = []), walls."""
        return np.concatenate([self.agent_pos, self.landmark_pos, self.obstacle_pos, self.wall_pos()], axis=1)


def _mean_std(x):
    """np.mean / np.std (population) along the last axis -- navigation_graph.py:617-621, 924-925."""
    return np.mean(x, axis=-1), np.std(x, axis=-1)


def decode_actions(cfg, actions):
    """environment.py:265-311: u = 5 * [a1 - a2, a3 - a4]; integer input = one-hot index."""
    a = np.asarray(actions)
    if a.ndim == 2:  # (n, N) indices
        oh = np.zeros(a.shape + (5,))
        np.put_along_axis(oh, a[..., None].astype(np.int64), 1.0, axis=-1)
        a = oh
    a = a.astype(np.float64)
    return SENSITIVITY * np.stack([a[..., 1] - a[..., 2], a[..., 3] - a[..., 4]], axis=-1)


def softplus_pen(d, dmin, k):
    """core.py:391 / :439: np.logaddexp(0, -(dist - dist_min)/k) * k."""
    return np.logaddexp(0, -(d - dmin) / k) * k


def environment_forces(cfg, st):
    """core.py:301-335 + :370-404 + :407-462 -> force on every agent, (n, N, 2).

    Pairs that contribute (core.py:373-376): agent-agent, agent-obstacle, agent-wall entity
    (a sphere of 'size' = width 0.1 at the wall's p_pos); landmarks have collide=False.
    d_min = size_a + size_b (cached branch, core.py:379-382).  status is never True here.
    """
    n, N = st.agent_pos.shape[:2]
    ap = st.agent_pos
    F = np.zeros((n, N, 2))
    others = [(ap, 2 * ENTITY_SIZE, True), (st.obstacle_pos, 2 * ENTITY_SIZE, False),
              (st.wall_pos(), ENTITY_SIZE + WALL_WIDTH, False)]
    for pos_b, dmin, is_agents in others:
        if pos_b.shape[1] == 0:
            continue
        delta = ap[:, :, None, :] - pos_b[:, None, :, :]  # (n, N, M, 2) = x_a - x_b
        d = np.sqrt(np.sum(np.square(delta), axis=-1))
        if is_agents:
            eye = np.eye(N, dtype=bool)[None]
            d = np.where(eye, 1.0, d)
        pen = softplus_pen(d, dmin, CONTACT_MARGIN)
        f = CONTACT_FORCE * delta / d[..., None] * pen[..., None]
        if is_agents:
            f = np.where(eye[..., None], 0.0, f)
        F += f.sum(axis=2)
    # walls proper: core.py:407-462
    for w in range(cfg.W):
        horiz = (st.wall_orient[:, w] == 0)[:, None]
        p_par = np.where(horiz, ap[:, :, 0], ap[:, :, 1])
        p_perp = np.where(horiz, ap[:, :, 1], ap[:, :, 0])
        e0 = st.wall_e0[:, w][:, None]; e1 = st.wall_e1[:, w][:, None]
        axis = st.wall_axis[:, w][:, None]
        s = ENTITY_SIZE
        beyond = (p_par < e0 - s) | (p_par > e1 + s)
        partial = (p_par < e0) | (p_par > e1)
        past = np.where(p_par < e0, p_par - e0, p_par - e1)
        past = np.where(partial & ~beyond, past, 0.0)
        theta = np.arcsin(past / s)
        dmin = np.cos(theta) * s + 0.5 * WALL_WIDTH
        dpos = p_perp - axis
        d = np.abs(dpos)
        pen = softplus_pen(d, dmin, WALL_CONTACT_MARGIN)
        with np.errstate(divide='ignore', invalid='ignore'):
            fmag = WALL_CONTACT_FORCE * dpos / d * pen
        f_perp = np.where(beyond, 0.0, np.cos(theta) * fmag)
        f_par = np.where(beyond, 0.0, np.sin(theta) * np.abs(fmag))
        fx = np.where(horiz, f_par, f_perp)
        fy = np.where(horiz, f_perp, f_par)
        F += np.stack([fx, fy], axis=-1)
    return F


def integrate(cfg, st, F):
    """core.py:338-356 (mass 1, damping .25, dt .1, speed clamp, p_dist, time)."""
    v = st.agent_vel * (1 - DAMPING)
    v = v + (F / 1.0) * DT
    if cfg.max_speed is not None:
        speed = np.sqrt(np.square(v[..., 0]) + np.square(v[..., 1]))
        over = speed > cfg.max_speed
        safe = np.where(over, speed, 1.0)
        v = np.where(over[..., None], v / safe[..., None] * cfg.max_speed, v)
    st.agent_vel = v
    st.agent_pos = st.agent_pos + v * DT
    st.p_dist = st.p_dist + np.sqrt(np.sum(np.square(v * DT), axis=-1))
    st.time = st.time + DT


def world_step(cfg, st, u):
    """core.py:250-274 World.step(): action force, environment force, integrate."""
    F = u * 1.0 + environment_forces(cfg, st)  # core.py:277-298 then :301-335
    integrate(cfg, st, F)


def distance_matrix(st):
    """core.py:204-228 calculate_distances -> cached_dist_mag (n, E, E)."""
    ep = st.entity_pos()
    delta = ep[:, :, None, :] - ep[:, None, :, :]
    return np.sqrt(np.sum(np.square(delta), axis=-1))


def obstacle_hit(cfg, st, pos):
    """navigation_graph.py:650-684 is_obstacle_collision for positions ``pos`` (n, K, 2), size .05."""
    n, K = pos.shape[:2]
    hit = np.zeros((n, K), dtype=bool)
    if cfg.O:
        d = np.sqrt(np.sum(np.square(st.obstacle_pos[:, None, :, :] - pos[:, :, None, :]), axis=-1))
        hit |= (d < 1.05 * (ENTITY_SIZE + ENTITY_SIZE)).any(axis=-1)
    s = ENTITY_SIZE
    for w in range(cfg.W):
        horiz = (st.wall_orient[:, w] == 0)[:, None]
        axis = st.wall_axis[:, w][:, None]; e0 = st.wall_e0[:, w][:, None]; e1 = st.wall_e1[:, w][:, None]
        p_perp = np.where(horiz, pos[:, :, 1], pos[:, :, 0])
        p_par = np.where(horiz, pos[:, :, 0], pos[:, :, 1])
        hit |= ((1.05 * (axis - s / 2) <= p_perp) & (p_perp <= 1.05 * (axis + s / 2)) &
                (1.05 * (e0 - s / 2) <= p_par) & (p_par <= 1.05 * (e1 + s / 2)))
    return hit


def fairness_from(vec):
    """mean/(std+1e-4) -- navigation_graph.py:766, :769, :851, :854."""
    m, s = _mean_std(vec)
    return m / (s + 0.0001)


def node_features(cfg, st):
    """navigation_graph.py:941-1035 + :1079-1124 -> (n, N, E, 11) relative node features;
    graph_feat_type 'global' (:981-1009, :1058-1077): (n, N, E, 7) [vel, pos, goal, type], the same for every ego."""
    n, N = st.agent_pos.shape[:2]
    L, O, W, E = cfg.L, cfg.O, cfg.W, cfg.E
    ep = st.entity_pos()
    ev = np.zeros((n, E, 2)); ev[:, :N] = st.agent_vel
    if cfg.graph_feat_type == 'global':
        if W:
            raise ValueError('wall not supported')   # :1075: _get_entity_feat_global knows no walls
        row = np.zeros((n, E, 7))
        row[..., 0:2] = ev
        row[..., 2:4] = ep
        row[..., 4:6] = ep
        row[:, :N, 4:6] = np.take_along_axis(st.landmark_pos, st.goal_match[..., None], axis=1)
        row[..., 6] = np.concatenate([np.zeros(N), np.ones(L), 2 * np.ones(O)])
        return np.broadcast_to(row[:, None], (n, N, E, 7)).copy()
    ego_p = st.agent_pos[:, :, None, :]; ego_v = st.agent_vel[:, :, None, :]
    rel_pos = ep[:, None, :, :] - ego_p
    rel_vel = ev[:, None, :, :] - ego_v
    out = np.zeros((n, N, E, 11))
    out[..., 0:2] = rel_vel
    out[..., 2:4] = rel_pos
    out[..., 4:6] = rel_pos
    out[..., 6:8] = rel_pos
    out[..., 8:10] = rel_pos
    goal = np.take_along_axis(st.landmark_pos, st.goal_match[..., None], axis=1)  # goal of entity-agent e
    out[:, :, :N, 4:6] = goal[:, None, :, :] - ego_p
    etype = np.concatenate([np.zeros(N), np.ones(L), 2 * np.ones(O), 3 * np.ones(W)])
    out[..., 10] = etype
    if W:
        oc = np.stack([st.wall_e0, st.wall_axis + WALL_WIDTH / 2], axis=-1)  # :1115 (orientation ignored)
        dc = np.stack([st.wall_e1, st.wall_axis - WALL_WIDTH / 2], axis=-1)  # :1116
        out[:, :, E - W:, 6:8] = oc[:, None, :, :] - ego_p
        out[:, :, E - W:, 8:10] = dc[:, None, :, :] - ego_p
    return out


def edge_list(cfg, dist):
    """navigation_graph.py:1037-1056 update_graph for ONE env: COO edges 0 < d <= max_edge_dist."""
    connect = (dist <= cfg.max_edge_dist) & (dist > 0)
    row, col = np.nonzero(connect)  # row-major, like csr->coo
    return np.stack([row, col]), dist[row, col]


def observe(cfg, st, fairness):
    """navigation_graph.py:826-857: [vel, pos, goal - pos, fairness] (n, N, 7)."""
    goal = np.take_along_axis(st.landmark_pos, st.goal_match[..., None], axis=1)
    return np.concatenate([st.agent_vel, st.agent_pos, goal - st.agent_pos, fairness[..., None]], axis=-1)


def env_step(cfg, st, actions):
    """multiagent/environment.py:816-877 MultiAgentGraphEnv.step for ``n`` envs at once.

    The reference walks agents i = 0..N-1 sequentially (obs, id, reward, graph, done, info) and
    ``info_callback`` mutates per-world vectors that the next agent's obs/reward read; that is
    restated in closed form: agent i sees ``dists_to_goal`` with entries j < i already refreshed
    this step and entries j >= i from the previous step.
    Returns dict of arrays; ``st`` is advanced in place.
    """
    n, N = st.agent_pos.shape[:2]
    st.cur_step = st.cur_step + 1  # environment.py:819 and :823 (current_step, current_time_step)
    u = decode_actions(cfg, actions)
    world_step(cfg, st, u)
    dist = distance_matrix(st)

    goal = np.take_along_axis(st.landmark_pos, st.goal_match[..., None], axis=1)
    dg = np.sqrt(np.sum(np.square(st.agent_pos - goal), axis=-1))  # :583, :774
    Dg_old, Tr_old = st.dists_to_goal, st.times_required
    open_ = (Tr_old == -1)
    arrive = (dg < cfg.min_dist_thresh) & open_  # :587
    Tr_new = np.where(arrive, (st.cur_step * DT)[:, None], Tr_old)  # :589
    Dg_new = np.where(open_, st.p_dist, Dg_old)  # :590, :597
    left_new = np.where(open_, dg, st.dist_left)  # :591, :598

    j = np.arange(N)
    lt = (j[None, :] < j[:, None])[None]  # [i, j] : j < i  -> fresh before agent i's obs/reward
    le = (j[None, :] <= j[:, None])[None]  # j <= i -> fresh after agent i's info
    mixed_pre = np.where(lt, Dg_new[:, None, :], Dg_old[:, None, :])  # (n, i, j)
    f_stale = fairness_from(mixed_pre)  # :769, :854
    f_fresh = fairness_from(st.p_dist)[:, None]  # :764-766, :849-851, :914-927
    fairness = np.where(Dg_old == -1, f_fresh, f_stale)

    # collisions: :701-705 (agent-agent, 1.05*(s+s)) and :650-684 (obstacles / wall boxes)
    dag = dist[:, :N, :N]
    ag_hits = ((dag < 1.05 * 2 * ENTITY_SIZE) & ~np.eye(N, dtype=bool)[None]).sum(axis=-1)
    ob_hit = obstacle_hit(cfg, st, st.agent_pos)

    # reward :760-824
    rew = np.where(dg < cfg.min_dist_thresh, float(cfg.goal_rew), -dg)
    rew = rew - cfg.collision_rew * ag_hits - cfg.collision_rew * ob_hit
    fr = cfg.fair_rew * np.tanh(fairness - cfg.zeroshift)
    fr = np.where(fr < -2, -2.0, fr)
    rew = np.clip(rew + fr, -2 * cfg.collision_rew, cfg.goal_rew + cfg.fair_rew)

    obs = observe(cfg, st, fairness)
    node = node_features(cfg, st)
    done = np.broadcast_to((st.cur_step >= cfg.episode_length)[:, None], (n, N)).copy()  # environment.py:237-247

    # info :577-647 (stats after agent i's own update)
    st.num_obst_coll = st.num_obst_coll + ob_hit
    st.num_agent_coll = st.num_agent_coll + ag_hits
    mixed_post = np.where(le, Dg_new[:, None, :], Dg_old[:, None, :])
    d_mean, d_std = _mean_std(mixed_post)
    t_mixed = np.where(le, Tr_new[:, None, :], Tr_old[:, None, :])
    t_mean, t_std = _mean_std(t_mixed)
    st.dists_to_goal, st.times_required, st.dist_left = Dg_new, Tr_new, left_new
    info = np.stack([left_new, Tr_new, st.num_agent_coll, st.num_obst_coll, d_mean, d_std,
                     d_mean / (d_std + 0.0001), Dg_new, st.time, t_mean, t_std,
                     t_mean / (t_std + 0.0001), st.min_time, rew], axis=-1)
    if cfg.collaborative:  # environment.py:867-870: [[sum]] * N
        rew_out = np.broadcast_to(rew.sum(axis=-1)[:, None, None], (n, N, 1)).copy()
    else:
        rew_out = rew
    return dict(obs=obs, node_obs=node, adj=dist, reward=rew_out, done=done, info=info)


def observe_reset(cfg, st):
    """environment.py:882-898 reset(): obs / node_obs / adj of a freshly reset world.
    All dists_to_goal are -1, so fairness = mean(p_dist)/(std(p_dist)+1e-4) = 0."""
    fairness = np.where(st.dists_to_goal == -1, fairness_from(st.p_dist)[:, None],
                        fairness_from(st.dists_to_goal)[:, None])
    return dict(obs=observe(cfg, st, fairness), node_obs=node_features(cfg, st), adj=distance_matrix(st))


# ----------------------------------------------------------------------------- reset

class NumpyGlobalStream:
    """Draws from NumPy's process-global MT19937 in the reference's call order
    (environment.py:192-196; navigation_graph.py:272, :288, :298, :394, :492)."""

    def uniform_pair(self, lo, hi):
        return np.random.uniform(lo, hi, 2)

    def uniform(self, lo, hi):
        return np.random.uniform(lo, hi)

    def choice_hv(self):
        return str(np.random.choice(['H', 'V']))


MAX_TRIES = 10000  # the reference loops until success; bounded here (and on the device)


def draw_wall_length(cfg, rng):
    """navigation_graph.py:183-185 (drawn once per env in make_world)."""
    return rng.uniform(0.2, 0.8) * cfg.world_size / 4


def _hit_one(cfg, st, e, p):
    return bool(obstacle_hit(cfg, _View(st, e), p[None, None, :])[0, 0])


class _View:
    """Single-env view with a leading batch axis of 1 (for the vectorised helpers)."""

    def __init__(self, st, e):
        for k in State.FIELDS:
            setattr(self, k, getattr(st, k)[e:e + 1])
    wall_pos = State.wall_pos
    entity_pos = State.entity_pos


def reset_env(cfg, st, e, rng, assign=lexifair):
    """navigation_graph.py:212-575 reset_world + random_scenario for env ``e`` (in place)."""
    N, L, O, W = cfg.N, cfg.L, cfg.O, cfg.W
    ws = cfg.world_size
    st.cur_step[e] = 0
    st.times_required[e] = -1; st.dists_to_goal[e] = -1; st.dist_left[e] = -1  # :217-221
    st.num_obst_coll[e] = 0; st.num_agent_coll[e] = 0  # :223-225
    st.p_dist[e] = 0; st.time[e] = 0  # :239-240
    for o in range(O):  # :271-275
        st.obstacle_pos[e, o] = 0.8 * rng.uniform_pair(-ws / 2, ws / 2)
    wall_position = rng.uniform(0.2, 0.9)  # :288 (drawn even when there are no walls)
    wall_axis = [wall_position * ws / 2, -wall_position * ws / 2]
    for w in range(W):  # :294-324
        st.wall_orient[e, w] = 0 if rng.choice_hv() == 'H' else 1
        st.wall_e0[e, w] = -st.wall_length[e]; st.wall_e1[e, w] = st.wall_length[e]
        st.wall_axis[e, w] = wall_axis[w]
    thr = 1.05 * 2 * ENTITY_SIZE
    k = 0; tries = 0
    while k < N:  # :389-457
        p = rng.uniform_pair(-ws / 2, ws / 2)
        tries += 1
        bad = _hit_one(cfg, st, e, p)
        if not bad and k:
            bad = bool((np.sqrt(np.sum(np.square(st.agent_pos[e, :k] - p), axis=-1)) < thr).any())  # :689-698
        if not bad or tries >= MAX_TRIES:
            st.agent_pos[e, k] = p; st.agent_vel[e, k] = 0
            k += 1; tries = 0
    k = 0; tries = 0
    while k < L:  # :472-535
        p = 0.8 * rng.uniform_pair(-ws / 2, ws / 2)
        tries += 1
        bad = _hit_one(cfg, st, e, p)
        if not bad and k:
            bad = bool((np.sqrt(np.sum(np.square(st.landmark_pos[e, :k] - p), axis=-1)) < thr).any())  # :707-716
        if not bad or tries >= MAX_TRIES:
            st.landmark_pos[e, k] = p
            k += 1; tries = 0
    if cfg.max_speed is not None:  # :545-547 -- uses the PREVIOUS episode's goal_match_index
        g = st.landmark_pos[e][st.goal_match[e]]
        st.min_time[e] = np.sqrt(np.sum(np.square(st.agent_pos[e] - g), axis=-1)) / cfg.max_speed
    costs = cost_matrix(st.agent_pos[e], st.landmark_pos[e])  # :555
    perm = assign(costs)  # :557-561
    if perm is None or np.shape(perm) != (N,):
        perm = np.arange(N)
    st.goal_match[e] = perm


def cost_matrix(agent_pos, goal_pos):
    """navigation_graph.py:555: scipy ``cdist(agent_pos, goal_pos)`` (Euclidean) -> (N, L)."""
    d = agent_pos[:, None, :] - goal_pos[None, :, :]
    return np.sqrt(np.sum(np.square(d), axis=-1))


class OracleGraphVecEnv:
    """Restates GraphMPEEnv construction (MPE_env.py:55-77) + the auto-reset wrappers
    (onpolicy/envs/env_wrappers.py:850-1026) for ``n`` envs, NumPy float64.

    ``mode='dummy'``: all envs share NumPy's global stream like GraphDummyVecEnv in one
    process (construct env r: make_world draws wall_length + a full reset, then
    ``np.random.seed(seed + 1000 r)``); ``step`` returns the 8-tuple with ``reset_count``.
    ``mode='subproc'``: every env owns a stream (its own process in the reference); with the
    NumPy backend that is emulated by saving/restoring the global state per env.
    ``streams``: optional factory ``(env, episode) -> stream`` (e.g. Philox) replacing NumPy.
    """

    def __init__(self, cfg, n, seeds=None, mode='dummy', streams=None, assign=lexifair):
        self.cfg, self.n, self.mode, self.streams, self.assign = cfg, n, mode, streams, assign
        self.st = State(cfg, n)
        self.episode = np.zeros(n, dtype=np.int64)
        self._np_states = [None] * n
        for e in range(n):
            rng = self._rng(e)
            self.st.wall_length[e] = draw_wall_length(cfg, rng)  # make_world
            reset_env(cfg, self.st, e, rng, assign)  # make_world -> reset_world (:209)
            self.episode[e] += 1
            if streams is None and seeds is not None:
                np.random.seed(int(seeds[e]))  # env.seed(...) train_mpe.py:31
            self._save(e)

    def _rng(self, e):
        if self.streams is not None:
            return self.streams(e, int(self.episode[e]))
        if self.mode == 'subproc' and self._np_states[e] is not None:
            np.random.set_state(self._np_states[e])
        return NumpyGlobalStream()

    def _save(self, e):
        if self.streams is None and self.mode == 'subproc':
            self._np_states[e] = np.random.get_state()

    def _reset_one(self, e):
        reset_env(self.cfg, self.st, e, self._rng(e), self.assign)
        self.episode[e] += 1
        self._save(e)

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
            for e in np.nonzero(done_all)[0]:
                reset_count = 1
                self._reset_one(e)
            o = observe_reset(self.cfg, self.st)
            for k in ('obs', 'node_obs', 'adj'):
                out[k] = np.where(done_all.reshape((-1,) + (1,) * (out[k].ndim - 1)), o[k], out[k])
        N = self.cfg.N
        adj = np.broadcast_to(out['adj'][:, None], (self.n, N) + out['adj'].shape[1:]).copy()
        res = (out['obs'], self._agent_id(), out['node_obs'], adj, out['reward'], out['done'], out['info'])
        if self.mode == 'dummy':
            return res + (reset_count,)
        return res
