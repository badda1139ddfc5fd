"""Vectorised-env API of the reference (onpolicy/envs/env_wrappers.py) over the HIP engine.

Same class names, constructor signature (a list of env factories), spaces attributes, return
arities and dtypes as the reference, so ``onpolicy``'s rMAPPO runner drops in unchanged:

* ``GraphSubprocVecEnv.step`` -> 7-tuple ``(obs, agent_id, node_obs, adj, rewards, dones, infos)``
  (reference env_wrappers.py:988-996), ``GraphDummyVecEnv.step`` -> 8-tuple with ``reset_count``
  (:912-928); ``reset`` -> 4-tuple (:997-1002).  NumPy float64 / int64 / bool like the reference.
* ``SubprocVecEnv`` / ``DummyVecEnv``: the non-graph twins, ``(obs, rews, dones, infos)`` and
  ``reset() -> obs`` (:272-282, :697-716).
* auto-reset when all agents of an env are done (:859-865) happens on the device.

There are no worker processes: every env lives in one ``RolloutEngine`` on one GPU.  For
throughput work use the engine directly (``venv.engine``): these wrappers copy every output to
the host each step, which only makes sense for small ``n_rollout_threads``.
"""
from abc import ABC, abstractmethod

import numpy as np

from .engine import RolloutEngine
from .infos import LazyInfos
from .MPE_env import EnvSpec


class ShareVecEnv(ABC):
    """reference env_wrappers.py:28-139"""
    closed = False
    viewer = None
    metadata = {'render.modes': ['human', 'rgb_array']}

    def __init__(self, num_envs, observation_space, share_observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.share_observation_space = share_observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def step_async(self, actions):
        pass

    @abstractmethod
    def step_wait(self):
        pass

    def close_extras(self):
        pass

    def close(self):
        if self.closed:
            return
        self.close_extras()
        self.closed = True

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def render(self, mode='human'):
        raise NotImplementedError('rendering is outside the MI355X hot path')


class _EngineVecEnv(ShareVecEnv):
    def __init__(self, env_fns, device='cuda:0', emit_graph=True):
        specs = [fn() for fn in env_fns]
        if not specs or not all(isinstance(s, EnvSpec) for s in specs):
            raise TypeError('env_fns must return fair_marl_amd.GraphMPEEnv / MPEEnv specs')
        spec = specs[0]
        self.spec = spec
        seed = spec.seed_value if spec.seed_value is not None else 1
        self.engine = RolloutEngine(spec.cfg, len(specs), device=device, seed=seed, emit_graph=emit_graph)
        ShareVecEnv.__init__(self, len(specs), spec.observation_space, spec.share_observation_space, spec.action_space)
        self.node_observation_space = spec.node_observation_space
        self.adj_observation_space = spec.adj_observation_space
        self.edge_observation_space = spec.edge_observation_space
        self.agent_id_observation_space = spec.agent_id_observation_space
        self.share_agent_id_observation_space = spec.share_agent_id_observation_space
        self.actions = None

    def step_async(self, actions):
        self.actions = np.asarray(actions)

    def _step_device(self):
        obs, ids, node, adj, rew, done, info = self.engine.step(self.actions, auto_reset=True)
        self.actions = None
        if self.spec.cfg.collaborative:
            # reference multiagent/environment.py:867-870: every agent gets [sum of the env's rewards]
            # (a list of 1-element lists -> shape (n, N, 1)); infos keep the individual rewards
            rew = rew.sum(dim=1, keepdim=True).expand_as(rew).unsqueeze(-1)
        return obs, ids, node, adj, rew, done, info

    @staticmethod
    def _np(t, dtype):
        return t.detach().cpu().numpy().astype(dtype)

    def close_extras(self):
        self.engine.close()

    def reset_task(self):
        raise NotImplementedError  # reference worker: env.reset_task() does not exist for these scenarios


class GraphSubprocVecEnv(_EngineVecEnv):
    """reference env_wrappers.py:951-1026 (spaces argument accepted and ignored like the reference)."""

    def __init__(self, env_fns, spaces=None, device='cuda:0'):
        _EngineVecEnv.__init__(self, env_fns, device)

    def step_wait(self):
        obs, ids, node, adj, rew, done, info = self._step_device()
        return (self._np(obs, np.float64), self._np(ids, np.int64), self._np(node, np.float64),
                self._np(adj, np.float64), self._np(rew, np.float64), self._np(done, bool),
                LazyInfos(self._np(info, np.float64), self.spec.cfg.scenario_name))

    def reset(self):
        obs, ids, node, adj = self.engine.reset()
        return (self._np(obs, np.float64), self._np(ids, np.int64), self._np(node, np.float64),
                self._np(adj, np.float64))


class GraphDummyVecEnv(GraphSubprocVecEnv):
    """reference env_wrappers.py:895-948: same data plus ``reset_count`` as 8th item."""

    def __init__(self, env_fns, device='cuda:0'):
        _EngineVecEnv.__init__(self, env_fns, device)

    def step_wait(self):
        res = GraphSubprocVecEnv.step_wait(self)
        reset_count = 1 if bool(res[5].all(axis=1).any()) else 0
        return res + (reset_count,)


class SubprocVecEnv(_EngineVecEnv):
    """reference env_wrappers.py:242-307 (env_name == 'MPE': graph outputs dropped)."""

    def __init__(self, env_fns, spaces=None, device='cuda:0'):
        _EngineVecEnv.__init__(self, env_fns, device, emit_graph=False)

    def step_wait(self):
        obs, ids, node, adj, rew, done, info = self._step_device()
        return (self._np(obs, np.float64), self._np(rew, np.float64), self._np(done, bool),
                LazyInfos(self._np(info, np.float64), self.spec.cfg.scenario_name))

    def reset(self):
        obs, ids, node, adj = self.engine.reset()
        return self._np(obs, np.float64)


class DummyVecEnv(SubprocVecEnv):
    """reference env_wrappers.py:686-729"""

    def __init__(self, env_fns, device='cuda:0'):
        _EngineVecEnv.__init__(self, env_fns, device, emit_graph=False)
