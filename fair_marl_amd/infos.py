"""Lazy view of the dense info record array as the reference's ``infos`` structure:
a sequence over envs of a list over agents of a dict (reference
multiagent/custom_scenarios/navigation_graph.py:625-647, multiagent/environment.py:861,
consumed by onpolicy/runner/shared/base_runner.py:208-243).  Building 65 536 x 32 dicts per
step is impossible, so dicts are materialised only for the entries actually touched."""
import numpy as np

INFO_KEYS = ('Dist_to_goal', 'Time_req_to_goal', 'Num_agent_collisions', 'Num_obst_collisions',
             'Distance_mean', 'Distance_variance', 'Mean_by_variance', 'Dists_traveled', 'Time_taken',
             'Time_mean', 'Time_stddev', 'Time_mean_by_stddev', 'Min_time_to_goal', 'individual_reward')
# fair_graph_formation.py:484-499: no time statistics, 'Formation_dist' instead (record slot 9)
FORMATION_KEYS = {'Dist_to_goal': 0, 'Time_req_to_goal': 1, 'Num_agent_collisions': 2, 'Num_obst_collisions': 3,
                  'Distance_mean': 4, 'Distance_variance': 5, 'Mean_by_variance': 6, 'Dists_traveled': 7,
                  'Time_taken': 8, 'Formation_dist': 9, 'Min_time_to_goal': 12, 'individual_reward': 13}
NAV_KEYS = {k: i for i, k in enumerate(INFO_KEYS)}


def key_map(scenario_name):
    """dict key -> record slot, in the reference's dict order (environment.py:861: individual_reward first)."""
    m = FORMATION_KEYS if scenario_name == 'fair_graph_formation' else NAV_KEYS
    order = ['individual_reward'] + [k for k in m if k != 'individual_reward']
    return [(k, m[k]) for k in order]


class AgentInfos(object):
    """infos[e]: behaves like the reference's per-env list of per-agent dicts."""

    def __init__(self, rec, keys):
        self._rec, self._keys = rec, keys  # (N, K) float array, [(key, slot)]

    def __len__(self):
        return self._rec.shape[0]

    def __getitem__(self, a):
        row = self._rec[a]
        return {k: float(row[j]) for k, j in self._keys}

    def __iter__(self):
        return (self[a] for a in range(len(self)))


class LazyInfos(object):
    """infos: sequence over envs; ``.array`` gives the raw (n, N, 14) records."""

    def __init__(self, records, scenario_name='navigation_graph'):
        self.array = np.asarray(records)
        self._keys = key_map(scenario_name)

    def __len__(self):
        return self.array.shape[0]

    def __getitem__(self, e):
        return AgentInfos(self.array[e], self._keys)

    def __iter__(self):
        return (self[e] for e in range(len(self)))
