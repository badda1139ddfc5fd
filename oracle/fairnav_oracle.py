"""TEST INFRASTRUCTURE ONLY — NumPy float64 restatement (oracle) of the
``nav_fairassign_fairrew_formation_graph`` scenario (the scenario of the shipped
FA / FA+FR weights).  Never imported by the product.

Physics and the fairness-scalar rule are shared with ``nav_oracle``; what differs is restated from
``multiagent/custom_scenarios/nav_fairassign_fairrew_formation_graph.py`` (cited as ``nf:<line>``):
per-step lexicographic-fair re-assignment inside ``reward(agent 0)``, stop-on-goal ``status`` (done per
agent, velocity zeroed inside ``reward``, no agent-agent force afterwards: ``core.py:394-398``),
float-valued goal occupancy / goal history mutated by ``observation`` in agent order, 11-wide obs,
13-wide node features, leave / re-enter logic in ``info_callback``.

Like the formation oracle the per-agent pass is emulated sequentially per env in the reference's call
order (``environment.py:832-864``).  Parity status: pinned by ``tests/golden/fnav_*.npz``; the
assignment itself is "parity unpinned" (Gurobi absent, see ``oracle/lexifair.py``).
"""
from dataclasses import dataclass, fields

import numpy as np

from . import nav_oracle as no
from .lexifair import lexifair
from .nav_oracle import CONTACT_FORCE, CONTACT_MARGIN, DT, ENTITY_SIZE, WALL_WIDTH

INFO_KEYS = ('Dist_to_goal', 'Time_req_to_goal', 'Num_agent_collisions', 'Num_obst_collisions', 'Distance_mean',
             'Distance_variance', 'Mean_by_variance', 'Dists_traveled', 'Time_taken', 'Time_mean', 'Time_stddev',
             'Time_mean_by_stddev', 'Min_time_to_goal', 'individual_reward')  # nf:573-590


@dataclass
class Config(no.Config):
    scenario_name: str = 'nav_fairassign_fairrew_formation_graph'
    min_obs_dist: float = 0.5  # onpolicy/config.py:188

    @property
    def obs_dim(self): return 11
    @property
    def node_feat(self): return 13


class State(no.State):
    FIELDS = no.State.FIELDS + ('goal_occ', 'goal_history', 'goal_reached', 'status')

    def __init__(self, cfg, n):
        super().__init__(cfg, n)
        N = cfg.N
        self.goal_occ = np.zeros((n, N))            # scenario.landmark_poses_occupied
        self.goal_history = -np.ones((n, N))        # scenario.goal_history
        self.goal_reached = -np.ones((n, N))        # scenario.goal_reached
        self.status = np.zeros((n, N))              # agent.status (0 / 1)


def environment_forces(cfg, st):
    """core.py:301-335 with ``status``: an agent whose status is True receives no agent-agent force
    (core.py:394-398); obstacle / wall-entity / wall forces are unaffected."""
    F = no.environment_forces(cfg, st)
    n, N = st.agent_pos.shape[:2]
    ap = st.agent_pos
    delta = ap[:, :, None, :] - ap[:, None, :, :]
    d = np.sqrt(np.sum(np.square(delta), axis=-1))
    eye = np.eye(N, dtype=bool)[None]
    d = np.where(eye, 1.0, d)
    pen = no.softplus_pen(d, 2 * ENTITY_SIZE, CONTACT_MARGIN)
    f = CONTACT_FORCE * delta / d[..., None] * pen[..., None]
    f = np.where(eye[..., None], 0.0, f).sum(axis=2)
    return F - f * st.status[..., None]


def obstacle_hit(cfg, st, e, pos):
    """nf:592-613: obstacles at 2.0 (s + s); wall boxes padded by 1.5 s."""
    s = ENTITY_SIZE
    for o in range(cfg.O):
        if np.linalg.norm(st.obstacle_pos[e, o] - pos) < 2.0 * (s + s):
            return True
    for w in range(cfg.W):
        axis, e0, e1 = st.wall_axis[e, w], st.wall_e0[e, w], st.wall_e1[e, w]
        pperp, ppar = (pos[1], pos[0]) if st.wall_orient[e, w] == 0 else (pos[0], pos[1])
        if axis - 1.5 * s <= pperp <= axis + 1.5 * s and e0 - 1.5 * s <= ppar <= e1 + 1.5 * s:
            return True
    return False


class _EnvPass:
    def __init__(self, cfg, st, e):
        self.cfg, self.st, self.e = cfg, st, e

    # ---- nf:840-1000 observation
    def observation(self, i):
        cfg, st, e = self.cfg, self.st, self.e
        x, v = st.agent_pos[e, i], st.agent_vel[e, i]
        G, occ, hist = st.landmark_pos[e], st.goal_occ[e], st.goal_history[e]
        thr = cfg.min_dist_thresh
        d = np.array([np.linalg.norm(x - g) for g in G])
        order = np.argsort(d)
        second = G[order[1]]
        second_occ = occ[order[1]]          # read before this call mutates anything
        if d.min() < cfg.min_obs_dist:
            chosen = int(np.argmin(d))
            goal = G[chosen]
            for g in np.where(d < cfg.min_obs_dist)[0]:
                if occ[g] == 1.0:
                    prox = np.array([np.linalg.norm(G[g] - a) for a in st.agent_pos[e]])
                    if not np.any(prox < thr):
                        occ[g] = np.min(prox)
            if d.min() < thr:
                occ[chosen] = 1.0
                hist[chosen] = i
            else:
                prox = np.array([np.linalg.norm(goal - a) for a in st.agent_pos[e]])
                closest = np.min(prox)
                if occ[chosen] == 1.0:
                    if np.any(prox < thr):
                        free = np.where(occ != 1)[0]
                        k = int(np.argmin(np.linalg.norm(x - G[free], axis=1)))
                        goal = G[free[k]]
                        chosen = k                      # quirk: the COMPACT index is used below (nf:888-890, :901-902)
                    else:
                        occ[chosen] = 1.0 - closest
                else:
                    occ[chosen] = 1.0 - closest
            g_occ, g_hist = occ[chosen], hist[chosen]
        else:
            free = np.where(occ != 1)[0]
            if len(free):
                k = int(np.argmin(np.linalg.norm(x - G[free], axis=1)))
                goal, g_occ, g_hist = G[free[k]], occ[free[k]], hist[free[k]]
            else:
                goal = x
                occ[:] = 0
                g_hist, g_occ = hist[i], occ[i]
        return np.concatenate([v, x, goal - x, [g_occ], [g_hist], second - x, [second_occ]])

    # ---- nf:1222-1334 node features relative to ego i
    def graph_observation(self, i):
        cfg, st, e = self.cfg, self.st, self.e
        N, L, O, W = cfg.N, cfg.L, cfg.O, cfg.W
        xi, vi = st.agent_pos[e, i], st.agent_vel[e, i]
        G, occ, hist = st.landmark_pos[e], st.goal_occ[e], st.goal_history[e]
        rows = []
        for a in range(N):
            xa = st.agent_pos[e, a]
            d = np.array([np.linalg.norm(xa - g) for g in G])
            if d.min() < cfg.min_obs_dist:
                c = int(np.argmin(d))
                goal, g_hist, g_occ = G[c], hist[c], occ[c]
            else:
                free = np.where(occ != 1)[0]
                if len(free):
                    k = int(np.argmin(np.linalg.norm(xa - G[free], axis=1)))
                    goal, g_occ, g_hist = G[free[k]], occ[free[k]], hist[free[k]]
                else:
                    goal = xa
                    occ[:] = 0
                    g_occ, g_hist = occ[a], hist[a]
            rp = xa - xi
            rows.append(np.hstack([st.agent_vel[e, a] - vi, rp, goal - xi, [g_occ], [g_hist], rp, rp, 0]))
        for l in range(L):
            rp = G[l] - xi
            rows.append(np.hstack([-vi, rp, rp, [1], [l], rp, rp, 1]))
        for o in range(O):
            rp = st.obstacle_pos[e, o] - xi
            rows.append(np.hstack([-vi, rp, rp, [1], [0], rp, rp, 2]))   # obstacle.id is None -> 0 (nf:1313)
        wp = st.wall_pos()[e]
        for w in range(W):
            rp = wp[w] - xi
            oc = np.array([st.wall_e0[e, w], st.wall_axis[e, w] + WALL_WIDTH / 2]) - xi
            dc = np.array([st.wall_e1[e, w], st.wall_axis[e, w] - WALL_WIDTH / 2]) - xi
            rows.append(np.hstack([-vi, rp, rp, [1], [w], oc, dc, 3]))
        return np.array(rows)


def env_step(cfg, st, actions):
    """MultiAgentGraphEnv.step (environment.py:816-877) with this scenario's callbacks."""
    n, N = st.agent_pos.shape[:2]
    st.cur_step = st.cur_step + 1
    u = no.decode_actions(cfg, actions)
    F = u * 1.0 + environment_forces(cfg, st)
    no.integrate(cfg, st, F)
    obs = np.zeros((n, N, 11)); node = np.zeros((n, N, cfg.E, 13)); rew = np.zeros((n, N))
    done = np.zeros((n, N), dtype=bool); info = np.zeros((n, N, len(INFO_KEYS)))
    thr = cfg.min_dist_thresh
    dist_after = None
    for e in range(n):
        ps = _EnvPass(cfg, st, e)
        x = st.agent_pos[e]
        G = st.landmark_pos[e]
        Dg, Tr = st.dists_to_goal[e], st.times_required[e]
        d_mean, d_std = np.mean(Dg), np.std(Dg)
        t_mean, t_std = np.mean(Tr), np.std(Tr)
        for i in range(N):
            obs[e, i] = ps.observation(i)
            # ---- reward nf:691-803
            f = (np.mean(st.p_dist[e]) / (np.std(st.p_dist[e]) + 0.0001)) if Dg[i] == -1 else d_mean / (d_std + 0.0001)
            if i == 0:  # nf:704-721 per-step fair re-assignment
                st.goal_match[e] = lexifair(no.cost_matrix(x, G))
            dgoal = np.linalg.norm(x[i] - G[st.goal_match[e, i]])
            r = 0.0
            if dgoal < thr:
                if st.status[e, i] == 0:
                    st.status[e, i] = 1
                    st.agent_vel[e, i] = 0.0          # nf:736-737: velocity zeroed inside reward
                    r += cfg.goal_rew
            else:
                r -= dgoal
            ag_hits = sum(1 for a in range(N) if a != i and np.linalg.norm(x[a] - x[i]) < 1.05 * 2 * ENTITY_SIZE)
            ob_hit = obstacle_hit(cfg, st, e, x[i])
            r = r - cfg.collision_rew * ag_hits - (cfg.collision_rew if ob_hit else 0)
            fr = cfg.fair_rew * np.tanh(f - cfg.zeroshift)
            if fr < -cfg.fair_rew:
                fr = -cfg.fair_rew
            r = float(np.clip(r + fr, -2 * cfg.collision_rew, cfg.goal_rew + cfg.fair_rew))
            rew[e, i] = r
            node[e, i] = ps.graph_observation(i)
            done[e, i] = bool(st.status[e, i]) or st.cur_step[e] >= cfg.episode_length   # environment.py:237-247
            # ---- info nf:489-590
            dl = np.array([np.linalg.norm(x[i] - g) for g in G])
            near = int(np.argmin(dl)); dn = dl[near]
            gr = st.goal_reached[e]
            if dn < thr and (near != gr[i] and gr[i] != -1):
                gr[i] = near; st.dist_left[e, i] = dn
            if dn < thr and Tr[i] == -1:
                Tr[i] = st.cur_step[e] * DT; Dg[i] = st.p_dist[e, i]; st.dist_left[e, i] = dn; gr[i] = near
            if Tr[i] == -1:
                Dg[i] = st.p_dist[e, i]; st.dist_left[e, i] = dn
            if dn > thr and Tr[i] != -1:
                Dg[i] = st.p_dist[e, i]; Tr[i] = st.cur_step[e] * DT; st.dist_left[e, i] = dn
            if dn < thr and near == gr[i]:
                st.dist_left[e, i] = dn; gr[i] = near
            if ob_hit:
                st.num_obst_coll[e, i] += 1
            st.num_agent_coll[e, i] += ag_hits
            d_mean, d_std = np.mean(Dg), np.std(Dg)
            t_mean, t_std = np.mean(Tr), np.std(Tr)
            info[e, i] = [st.dist_left[e, i], Tr[i], st.num_agent_coll[e, i], st.num_obst_coll[e, i], d_mean, d_std,
                          d_mean / (d_std + 0.0001), Dg[i], Tr[i], t_mean, t_std, t_mean / (t_std + 0.0001),
                          st.min_time[e, i], r]
    # NB: reward() may have zeroed velocities, which only matters for later rows / the next step; the
    # distance matrix depends on positions only
    dist_after = no.distance_matrix(st)
    return dict(obs=obs, node_obs=node, adj=dist_after, reward=rew, done=done, info=info)


def observe_reset(cfg, st, envs=None):
    n, N = st.agent_pos.shape[:2]
    obs = np.zeros((n, N, 11)); node = np.zeros((n, N, cfg.E, 13))
    for e in (range(n) if envs is None else envs):
        ps = _EnvPass(cfg, st, e)
        for i in range(N):
            obs[e, i] = ps.observation(i)
            node[e, i] = ps.graph_observation(i)
    return dict(obs=obs, node_obs=node, adj=no.distance_matrix(st))


def reset_env(cfg, st, e, rng, assign=lexifair):
    """nf:212-262 reset_world + nf:264-480 random_scenario."""
    N, L, O, W = cfg.N, cfg.L, cfg.O, cfg.W
    ws, s = cfg.world_size, ENTITY_SIZE
    st.cur_step[e] = 0
    st.times_required[e] = -1; st.dists_to_goal[e] = -1; st.dist_left[e] = -1
    st.num_obst_coll[e] = 0; st.num_agent_coll[e] = 0
    st.goal_match[e] = np.arange(N); st.goal_history[e] = -1; st.goal_reached[e] = -1
    st.wall_length[e] = rng.uniform(0.2, 0.8) * ws / 4      # nf:239-241: re-drawn at every reset
    st.p_dist[e] = 0; st.time[e] = 0
    for o in range(O):
        st.obstacle_pos[e, o] = 0.8 * rng.uniform_pair(-ws / 2, ws / 2)
    wall_position = rng.uniform(0.2, 0.9)
    wall_axis = [wall_position * ws / 2, -wall_position * ws / 2]
    for w in range(W):
        st.wall_orient[e, w] = 0 if rng.choice_hv() == 'H' else 1
        st.wall_e0[e, w] = -st.wall_length[e]; st.wall_e1[e, w] = st.wall_length[e]
        st.wall_axis[e, w] = wall_axis[w]
    k = tries = 0
    while k < N:
        p = rng.uniform_pair(-ws / 2, ws / 2)
        tries += 1
        bad = obstacle_hit(cfg, st, e, p)
        if not bad and k:
            bad = bool((np.sqrt(np.sum(np.square(st.agent_pos[e, :k] - p), axis=-1)) < 1.05 * 2 * s).any())
        if not bad or tries >= no.MAX_TRIES:
            st.agent_pos[e, k] = p; st.agent_vel[e, k] = 0; st.status[e, k] = 0
            k += 1; tries = 0
    k = tries = 0
    while k < L:
        p = 0.8 * rng.uniform_pair(-ws / 2, ws / 2)
        tries += 1
        bad = obstacle_hit(cfg, st, e, p)
        if not bad and k:
            bad = bool((np.sqrt(np.sum(np.square(st.landmark_pos[e, :k] - p), axis=-1)) < 1.2 * 2 * s).any())   # nf:643
        if not bad or tries >= no.MAX_TRIES:
            st.landmark_pos[e, k] = p
            k += 1; tries = 0
    st.goal_occ[e] = 0
    if cfg.max_speed is not None:   # goal_match_index was just reset to arange (nf:233)
        st.min_time[e] = np.sqrt(np.sum(np.square(st.agent_pos[e] - st.landmark_pos[e][:N]), axis=-1)) / cfg.max_speed
    st.goal_match[e] = assign(no.cost_matrix(st.agent_pos[e], st.landmark_pos[e]))


class OracleFairNavVecEnv:
    """GraphMPEEnv x n + auto-reset wrapper semantics (episodes end early once every agent has status)."""

    def __init__(self, cfg, n, seeds=None, mode='dummy', streams=None):
        self.cfg, self.n, self.mode, self.streams = cfg, n, mode, streams
        self.st = State(cfg, n)
        self.episode = np.zeros(n, dtype=np.int64)
        for e in range(n):
            rng = self._rng(e)
            rng.uniform(0.2, 0.4)                      # make_world's own wall_length draw (nf:183), overwritten
            reset_env(cfg, self.st, e, rng)
            self.episode[e] += 1
            if streams is None and seeds is not None:
                np.random.seed(int(seeds[e]))
        for e in range(n):   # env constructors call observation / graph_observation once (environment.py:112-117, :788-790)
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
            for k in ('obs', 'node_obs', 'adj'):
                out[k][idx] = o[k][idx]
        N = self.cfg.N
        adj = np.broadcast_to(out['adj'][:, None], (self.n, N) + out['adj'].shape[1:]).copy()
        res = (out['obs'], self._agent_id(), out['node_obs'], adj, out['reward'], out['done'], out['info'])
        return res + (reset_count,) if self.mode == 'dummy' else res
