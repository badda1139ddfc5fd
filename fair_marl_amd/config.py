"""Scenario configuration: the argparse knobs the reference scenarios read in ``make_world``
(reference multiagent/custom_scenarios/navigation_graph.py:94-129, 208; defaults from
onpolicy/config.py:176-252 and onpolicy/scripts/train_mpe.py:71-106)."""
from dataclasses import dataclass, fields

from . import _lib


@dataclass
class EnvConfig:
    scenario_name: str = 'navigation_graph'
    num_agents: int = 2
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
    min_obs_dist: float = 0.5   # nav_fairassign_fairrew_formation_graph only (onpolicy/config.py:188)

    def __post_init__(self):
        if self.scenario_name == 'fair_graph_formation':
            self.num_walls = 2  # hard-coded by the scenario (reference fair_graph_formation.py:184), args ignored

    @classmethod
    def from_args(cls, args):
        """Build from an argparse.Namespace (or any object / dict with these attributes)."""
        get = (lambda k: args[k]) if isinstance(args, dict) else (lambda k: getattr(args, k))
        has = (lambda k: k in args) if isinstance(args, dict) else (lambda k: hasattr(args, k))
        return cls(**{f.name: get(f.name) for f in fields(cls) if has(f.name)})

    def validate(self):
        if self.scenario_name not in _lib.SCENARIOS:
            raise NotImplementedError('scenario %r is outside the MI355X hot path (supported: %s)'
                                      % (self.scenario_name, ', '.join(_lib.SCENARIOS)))
        if self.graph_feat_type not in ('relative', 'global'):
            raise ValueError('graph_feat_type must be relative or global')
        if self.graph_feat_type == 'global' and self.scenario_name != 'navigation_graph':
            raise NotImplementedError("graph_feat_type='global' is built for navigation_graph only")
        if self.graph_feat_type == 'global' and self.num_walls:
            # the reference itself fails here: _get_entity_feat_global has no wall branch (navigation_graph.py:1075)
            raise ValueError('wall not supported with graph_feat_type=global')
        if self.num_scripted_agents:
            raise NotImplementedError('scripted agents are not part of the hot path')

    @property
    def N(self): return self.num_agents
    @property
    def E(self): return self.num_agents + self.num_landmarks + self.num_obstacles + self.num_walls
    @property
    def obs_dim(self): return {'navigation_graph': 7, 'fair_graph_formation': 6}.get(self.scenario_name, 11)
    @property
    def node_feat(self):
        if self.graph_feat_type == 'global':
            return 7
        return {'navigation_graph': 11, 'fair_graph_formation': 12}.get(self.scenario_name, 13)

    def to_c(self, n_envs, seed=0, env_offset=0, async_reset=False, envs_per_workgroup=0):
        c = _lib.FmarlConfig()
        c.flags = (_lib.FLAG_ASYNC_RESET if async_reset else 0) | (_lib.FLAG_GLOBAL_FEATURES if self.graph_feat_type == 'global' else 0)
        c.scenario = _lib.SCENARIOS[self.scenario_name]
        c.n_envs = int(n_envs)
        c.num_agents, c.num_landmarks = int(self.num_agents), int(self.num_landmarks)
        c.num_obstacles, c.num_walls = int(self.num_obstacles), int(self.num_walls)
        c.episode_length = int(self.episode_length)
        c.has_max_speed = 0 if self.max_speed is None else 1
        c.env_offset = int(env_offset)
        c.world_size = float(self.world_size)
        c.max_speed = 0.0 if self.max_speed is None else float(self.max_speed)
        c.collision_rew, c.goal_rew = float(self.collision_rew), float(self.goal_rew)
        c.min_dist_thresh, c.fair_rew = float(self.min_dist_thresh), float(self.fair_rew)
        c.zeroshift, c.max_edge_dist = float(self.zeroshift), float(self.max_edge_dist)
        c.min_obs_dist = float(self.min_obs_dist)
        c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        c.envs_per_workgroup = int(envs_per_workgroup)
        return c
