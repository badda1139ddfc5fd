"""Lazy view of the dense info record array as the reference's ``infos`` structure:
a sequence over envs of a list over agents of a dict (reference
multiagent/custom_scenarios/navigation_graph.py:625-647, multiagent/environment.py:861,
consumed by onpolicy/runner/shared/base_runner.py:208-243).  Building 65 536 x 32 dicts per
step is impossible, so dicts are materialised only for the entries actually touched."""
import numpy as np

INFO_KEYS = ('Dist_to_goal', 'Time_req_to_goal', 'Num_agent_collisions', 'Num_obst_collisions',
             'Distance_mean', 'Distance_variance', 'Mean_by_variance', 'Dists_traveled', 'Time_taken',
             'Time_mean', 'Time_stddev', 'Time_mean_by_stddev', 'Min_time_to_goal', 'individual_reward')
_ORDER = ('individual_reward',) + INFO_KEYS[:-1]   # environment.py:861: individual_reward first


class AgentInfos(object):
    """infos[e]: behaves like the reference's per-env list of per-agent dicts."""

    def __init__(self, rec):
        self._rec = rec  # (N, K) float array

    def __len__(self):
        return self._rec.shape[0]

    def __getitem__(self, a):
        row = self._rec[a]
        return {k: float(row[INFO_KEYS.index(k)]) for k in _ORDER}

    def __iter__(self):
        return (self[a] for a in range(len(self)))


class LazyInfos(object):
    """infos: sequence over envs; ``.array`` gives the raw (n, N, 14) records."""

    def __init__(self, records):
        self.array = np.asarray(records)

    def __len__(self):
        return self.array.shape[0]

    def __getitem__(self, e):
        return AgentInfos(self.array[e])

    def __iter__(self):
        return (self[e] for e in range(len(self)))
