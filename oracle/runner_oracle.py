"""TEST INFRASTRUCTURE ONLY — NumPy restatement (oracle) of the runner-side consumers of the env outputs
(SURVEY.md section 8 f-2, f-3, f-4 and a8).  Never imported by the product.

Parity status: **pinned** by ``tests/golden/runner_*.npz`` and ``edges_kat.npz`` — outputs of the reference's own
``GMPERunner.warmup`` / ``insert``, ``GraphReplayBuffer``, ``Runner.process_infos`` + ``get_*`` readers,
``Scenario.update_graph`` and ``TransformerConvNet.processAdj``, produced by ``tests/golden/gen_runner.py``
(``tests/test_runner_golden.py`` replays them).  Reference paths below are relative to the reference root.
"""
import numpy as np


class ReplayBuffer:
    """onpolicy/utils/graph_buffer.py:45-283 (env-facing arrays only) driven by the arithmetic of
    onpolicy/runner/shared/graph_mpe_runner.py:178-203 (warmup) and :438-488 (insert)."""

    def __init__(self, T, n, N, D, E, F):
        self.T, self.n, self.N = T, n, N
        f32 = np.float32
        self.share_obs = np.zeros((T + 1, n, N, N * D), f32)      # graph_buffer.py:84-88 (centralized V)
        self.obs = np.zeros((T + 1, n, N, D), f32)                # :89-93
        self.node_obs = np.zeros((T + 1, n, N, E, F), f32)        # :95-99
        self.adj = np.zeros((T + 1, n, N, E, E), f32)             # :100-104
        self.agent_id = np.zeros((T + 1, n, N, 1), np.int32)      # :105-109
        self.share_agent_id = np.zeros((T + 1, n, N, N), np.int32)  # :110-114
        self.rewards = np.zeros((T, n, N, 1), f32)                # :152-156
        self.masks = np.ones((T + 1, n, N, 1), f32)               # :158-161
        self.active_masks = np.ones_like(self.masks)              # :164
        self.step = 0

    def _share(self, x):   # graph_mpe_runner.py:470-478: (n, N, d) -> (n, N, N * d), every agent sees all
        flat = x.reshape(self.n, -1)
        return np.expand_dims(flat, 1).repeat(self.N, axis=1)

    def warmup(self, obs, agent_id, node_obs, adj):   # graph_mpe_runner.py:178-203
        self.share_obs[0] = self._share(obs); self.obs[0] = obs; self.node_obs[0] = node_obs; self.adj[0] = adj
        self.agent_id[0] = agent_id; self.share_agent_id[0] = self._share(agent_id)

    def insert(self, obs, agent_id, node_obs, adj, rewards, dones):
        """graph_mpe_runner.py:438-488 + graph_buffer.py:226-250; ``rewards`` (n, N), ``dones`` (n, N) bool."""
        dones_env = np.all(dones, axis=1)                     # :444
        masks = np.ones((self.n, self.N, 1), np.float32)
        masks[dones] = 0                                      # :452-458
        active = np.ones((self.n, self.N, 1), np.float32)
        active[dones] = 0                                     # :460-463
        active[dones_env] = 1                                 # :464-465
        t = self.step
        self.share_obs[t + 1] = self._share(obs); self.obs[t + 1] = obs; self.node_obs[t + 1] = node_obs
        self.adj[t + 1] = adj; self.agent_id[t + 1] = agent_id; self.share_agent_id[t + 1] = self._share(agent_id)
        self.rewards[t] = rewards[:, :, None]                 # graph_mpe_runner.py:80 / :121
        self.masks[t + 1] = masks; self.active_masks[t + 1] = active
        self.step = (t + 1) % self.T                          # graph_buffer.py:250

    def after_update(self):   # graph_buffer.py:252-283
        for k in ('share_obs', 'obs', 'node_obs', 'adj', 'agent_id', 'share_agent_id', 'masks', 'active_masks'):
            a = getattr(self, k)
            a[0] = a[-1]


def process_adj(adj, max_edge_dist):
    """onpolicy/algorithms/utils/gnn.py:307-326 on a float32 batch (B, E, E) (or one (E, E) matrix): edges with
    0 < adj < max_edge_dist (strict, float32 compare) in row-major order; node ids offset by b * E for a batch
    (:321-323 == the PyG collation of :243-253).  Returns (edge_index int64 (2, nnz), edge_attr f32 (nnz,))."""
    adj = np.asarray(adj, dtype=np.float32)
    connect = (adj < np.float32(max_edge_dist)) & (adj > 0)
    idx = np.nonzero(connect)
    attr = adj[idx]
    if adj.ndim == 3:
        off = idx[0] * adj.shape[-1]
        return np.stack([off + idx[1], off + idx[2]]).astype(np.int64), attr
    return np.stack(idx).astype(np.int64), attr


def update_graph(ent_pos, max_edge_dist):
    """multiagent/core.py:204-228 calculate_distances + navigation_graph.py:1037-1056 update_graph for one env:
    float64 distances, ``(d <= max_edge_dist) & (d > 0)``, COO in row-major order, weights = distances."""
    delta = ent_pos[:, None, :] - ent_pos[None, :, :]
    d = np.sqrt(delta[..., 0] * delta[..., 0] + delta[..., 1] * delta[..., 1])   # np.linalg.norm(axis=2)
    row, col = np.nonzero((d <= max_edge_dist) & (d > 0))
    return np.stack([row, col]), d[row, col]


# key of the info dict -> name in env_infos (base_runner.py:245-275)
ENV_INFO_NAMES = {
    'individual_reward': 'individual_rewards', 'Time_req_to_goal': 'time_to_goal', 'Min_time_to_goal': 'min_time_to_goal',
    'Dist_to_goal': 'dist_to_goal', 'Num_agent_collisions': 'num_agent_collisions', 'Num_obst_collisions': 'num_obstacle_collisions',
    'Distance_mean': 'distance_mean', 'Distance_variance': 'distance_variance', 'Mean_by_variance': 'mean_variance',
    'Dists_traveled': 'dists_traveled', 'Time_taken': 'time_taken', 'Formation_dist': 'formation_dist', 'Time_mean': 'time_mean',
    'Time_stddev': 'time_variance', 'Time_mean_by_stddev': 'time_mn_by_stddev'}


def process_infos(info, keys, episode_length, dt=0.1):
    """base_runner.py:197-276: ``info`` (n, N, K) with columns ``keys`` -> dict 'agent<i>/<name>' -> list over the
    envs; Time_req_to_goal == -1 becomes episode_length * dt (:212-215); names whose key the scenario does not emit
    map to empty lists."""
    n, N, _ = info.shape
    out = {}
    for a in range(N):
        for key, name in ENV_INFO_NAMES.items():
            if key in keys:
                v = info[:, a, list(keys).index(key)].astype(np.float64)
                if key == 'Time_req_to_goal':
                    v = np.where(v == -1, episode_length * dt, v)
                out['agent%d/%s' % (a, name)] = list(v)
            else:
                out['agent%d/%s' % (a, name)] = []
    return out


# base_runner.py:308-420: every reader returns [v[0] for the keys whose name contains the pattern], in dict order
METRIC_PATTERNS = {'get_fairness_metric': 'mean_variance', 'get_dist_mean': 'distance_mean', 'get_dist_std': 'distance_variance',
                   'get_time_fairness': 'time_mn_by_stddev', 'get_time_mean': 'time_mean', 'get_time_std': 'time_variance'}


def metric(env_infos, reader):
    pat = METRIC_PATTERNS[reader]
    return [v[0] for k, v in env_infos.items() if pat in k]


# ----------------------------------------------------------------------------------------------------------------------
# Learner side of the buffer (added in round 3): what the trainer computes on the filled buffer
# before the first gradient step.  Pinned by tests/golden/learner_*.npz (tests/golden/gen_learner.py: the reference's own
# compute_returns in all twelve branches, GR_MAPPO.train's advantages, both minibatch generators).

def compute_returns(rewards, value_preds, masks, bad_masks, next_value, gamma, gae_lambda, use_gae=True, proper=False, norm=None):
    """onpolicy/utils/graph_buffer.py:285-366.  ``rewards`` (T, ...), ``value_preds`` / ``masks`` / ``bad_masks`` (T + 1, ...)
    float32; ``norm`` = None or (mean, stddev) float32 of the value normaliser, whose ``denormalize`` is ``x * stddev + mean``
    in float32, product rounded before the sum (valuenorm.py:92-104, popart.py:101-111).  Float32 throughout, in the
    reference's order of operations (Python scalars enter as float32; gamma * gae_lambda is formed in double first).
    Returns (returns (T + 1, ...), value_preds with the last slot = next_value when use_gae)."""
    f32 = np.float32
    T = rewards.shape[0]
    v = value_preds.astype(f32).copy()
    ret = np.zeros_like(v)
    g, gl = f32(gamma), f32(float(gamma) * float(gae_lambda))
    if norm is None:
        dn = lambda x: x  # noqa: E731
    else:
        mean, std = f32(norm[0]), f32(norm[1])
        dn = lambda x: x * std + mean  # noqa: E731  (two float32 roundings)
    if use_gae:
        v[T] = next_value                                                       # :299 / :340
        gae = np.zeros_like(v[0])
        for t in range(T - 1, -1, -1):
            m1 = masks[t + 1]
            delta = rewards[t] + g * dn(v[t + 1]) * m1 - dn(v[t])              # :305-309 / :316-318 / :344-347 / :353-355
            if proper and norm is not None:
                gae = delta + gl * gae * m1                                     # :310-311
            else:
                gae = delta + gl * m1 * gae                                     # :319-320 / :348-349 / :356-357
            if proper:
                gae = gae * bad_masks[t + 1]                                    # :312 / :321
            ret[t] = gae + dn(v[t])                                             # :313-314 / :322 / :350-351 / :358
    else:
        ret[T] = next_value                                                     # :324 / :360
        for t in range(T - 1, -1, -1):
            acc = ret[t + 1] * g * masks[t + 1] + rewards[t]                    # :327-328 / :333-334 / :362-364
            if proper:
                b = bad_masks[t + 1]
                acc = acc * b + (f32(1) - b) * dn(v[t])                         # :329-331 / :335-337
            ret[t] = acc
    return ret, v


def advantages(returns, value_preds, active_masks, norm=None):
    """onpolicy/algorithms/graph_mappo.py:294-304: returns[:-1] - (denormalised) value_preds[:-1], then standardised by the
    mean / std over the entries whose active mask is not 0 (np.nanmean / np.nanstd of the float32 array), eps 1e-5."""
    f32 = np.float32
    v = value_preds[:-1]
    if norm is not None:
        v = v * f32(norm[1]) + f32(norm[0])
    adv = returns[:-1] - v
    keep = active_masks[:-1] != 0.0
    sel = adv[keep]
    mean = sel.mean(dtype=np.float32)
    std = np.sqrt(((sel - mean) ** 2).mean(dtype=np.float32))
    return (adv - mean) / (std + 1e-5)


def feed_forward_rows(perm, T, n, N, num_mini_batch):
    """graph_buffer.py:384-400: the (t, env, agent) of every row of every minibatch -- the flat index over (T, n, N) in C
    order is the permutation entry itself."""
    size = (T * n * N) // num_mini_batch
    out = []
    for b in range(num_mini_batch):
        idx = np.asarray(perm[b * size:(b + 1) * size], dtype=np.int64)
        out.append((idx // (n * N), (idx // N) % n, idx % N))
    return out


def recurrent_rows(perm, T, n, N, num_mini_batch, chunk):
    """graph_buffer.py:613-622, 673-700, 733-750: arrays are laid out (n, N, T) -> flat, cut into chunks of ``chunk``
    consecutive entries (a chunk may run over the end of one (env, agent) series into the next), a minibatch is
    ``data_chunks // num_mini_batch`` permuted chunks stacked on axis 1 and flattened (L, Nmb) -> rows l * Nmb + j.
    Returns per minibatch ((t, env, agent) of the rows, (t, env, agent) of the first entry of each chunk: the rnn states)."""
    chunks = (T * n * N) // chunk
    size = chunks // num_mini_batch
    out = []
    for b in range(num_mini_batch):
        c = np.asarray(perm[b * size:(b + 1) * size], dtype=np.int64)
        flat = (c[None, :] * chunk + np.arange(chunk)[:, None]).reshape(-1)      # row l * Nmb + j <- entry c_j * L + l
        split = lambda f: (f % T, f // (T * N), (f // T) % N)  # noqa: E731
        out.append((split(flat), split(c * chunk)))
    return out


def gather_minibatch(buf, adv, rows, first=None):
    """The 16 arrays a generator yields (graph_buffer.py:437-453 / :751-758) for rows (t, env, agent); ``buf`` = dict of the
    buffer's arrays (``adj_env`` (T + 1, n, E, E): one matrix per env, as the runner's insert stores N copies of)."""
    t, e, a = rows
    ft, fe, fa = rows if first is None else first
    share = buf['obs'].reshape(buf['obs'].shape[0], buf['obs'].shape[1], -1)
    aid = buf.get('available_actions')
    return (share[t, e], buf['obs'][t, e, a], buf['node_obs'][t, e, a], buf['adj_env'][t, e], buf['agent_id'][t, e, a],
            buf['share_agent_id'][t, e, a], buf['rnn_states'][ft, fe, fa], buf['rnn_states_critic'][ft, fe, fa],
            buf['actions'][t, e, a], buf['value_preds'][t, e, a], buf['returns'][t, e, a], buf['masks'][t, e, a],
            buf['active_masks'][t, e, a], buf['action_log_probs'][t, e, a], adv[t, e, a], None if aid is None else aid[t, e, a])
