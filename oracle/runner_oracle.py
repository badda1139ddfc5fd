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
