"""Factories with the reference's names (reference multiagent/MPE_env.py:21-77).

The reference builds one Python env object per call; here the call only records the arguments
(and, through ``seed``, the stream seed -- reference onpolicy/scripts/train_mpe.py:31) in an
``EnvSpec``.  The vec-env wrappers turn a list of such specs into ONE device-resident engine,
so ``GraphSubprocVecEnv([get_env_fn(i) for i in range(n)])`` keeps working unchanged."""
from .config import EnvConfig
from .spaces import Box, Discrete


class EnvSpec(object):
    def __init__(self, args, graph):
        self.args = args
        self.cfg = EnvConfig.from_args(args)
        self.cfg.validate()
        self.graph = graph
        self.seed_value = None
        c = self.cfg
        N, E = c.N, c.E
        self.n = self.num_agents = N
        # multiagent/environment.py:124-181, :781-813
        self.observation_space = [Box(shape=(c.obs_dim,)) for _ in range(N)]
        self.share_observation_space = [Box(shape=(c.obs_dim * N,)) for _ in range(N)]
        self.action_space = [Discrete(5) for _ in range(N)]
        self.node_observation_space = [Box(shape=(E, c.node_feat)) for _ in range(N)]
        self.adj_observation_space = [Box(shape=(E, E)) for _ in range(N)]
        self.edge_observation_space = [Box(shape=(1,)) for _ in range(N)]
        self.agent_id_observation_space = [Box(shape=(1,)) for _ in range(N)]
        self.share_agent_id_observation_space = [Box(shape=(N,)) for _ in range(N)]

    def seed(self, seed=None):
        self.seed_value = 1 if seed is None else int(seed)  # multiagent/environment.py:192-196

    def close(self):
        pass


def GraphMPEEnv(args):
    assert 'graph' in args.scenario_name, 'Only use graph env for graph scenarios'  # MPE_env.py:61
    return EnvSpec(args, graph=True)


def MPEEnv(args):
    # MultiAgentPPOEnv.step (multiagent/environment.py:675-704) never calls graph_observation.  In navigation_graph that
    # call has no side effect, so the same kernels serve with the graph emission off; in the two formation scenarios it
    # mutates the occupancy flags the next observation reads, so their non-graph form is a different environment.
    if args.scenario_name != 'navigation_graph':
        raise NotImplementedError('MPEEnv (non-graph) is built for navigation_graph only; %s changes behaviour without '
                                  'its graph_observation calls' % args.scenario_name)
    return EnvSpec(args, graph=False)
