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
import os
import warnings
from abc import ABC, abstractmethod

import numpy as np
import torch

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
    # adj as the reference shapes it, (n, N, E, E) float64 -- N identical copies of every env's matrix
    # (navigation_graph.py:1033).  By default the copies are a read-only stride-0 NumPy view of ONE (n, E, E) array: equal
    # values, shape and dtype, `adj.copy()` / indexing / np.concatenate as the runner uses them (graph_mpe_runner.py:200,
    # :404, graph_buffer.py:229) behave the same, but N times fewer bytes cross PCIe and the host (784 MB -> 24.5 MB per
    # step at 512 envs x 32 agents).  Set to True for N materialised, writable copies.
    materialize_adj = False
    # Ownership of the returned arrays.  0 (default): every array is fresh and the caller's for good, exactly like the arrays
    # the reference's workers pipe back.  k > 0 (opt-in): the float64 arrays of one call are handed out again k calls later
    # -- "valid until k steps later", for rollout loops that copy what they keep into their own buffers right away (the
    # reference runner's insert does: graph_buffer.py:84-165) and do not want 20 MB of first-touch page faults per step.
    reuse_outputs = 0

    def __init__(self, env_fns, device='cuda:0', emit_graph=True):
        specs = [fn() for fn in env_fns]
        if not specs or not all(isinstance(s, EnvSpec) for s in specs):
            raise TypeError('env_fns must return fair_marl_amd.GraphMPEEnv / MPEEnv specs')
        spec = specs[0]
        self.spec = spec
        seed = spec.seed_value if spec.seed_value is not None else 1
        # The reference seeds worker r with seed + 1000 r (onpolicy/scripts/train_mpe.py:31; eval: 50000 seed + 10000 r,
        # :56) and every worker owns a NumPy stream.  Here env r draws from the counter-based stream (seed of env 0, r):
        # one experiment seed plus the rank.  Seeds in such an arithmetic progression lose nothing; anything else
        # cannot be honoured per env.
        seeds = [s.seed_value for s in specs]
        if any(v is not None for v in seeds):
            filled = [1 if v is None else v for v in seeds]
            steps = {b - a for a, b in zip(filled, filled[1:])}
            if len(steps) > 1:
                warnings.warn('fair_marl_amd: per-env seeds %s are not of the form seed + r * const; env r draws from the '
                              'stream (seed=%d, env index r) -- only the first env\'s seed is used' % (filled[:4] + ['...'] if len(filled) > 4 else filled, seed),
                              UserWarning, stacklevel=3)
        self.engine = RolloutEngine(spec.cfg, len(specs), device=device, seed=seed, emit_graph=emit_graph)
        ShareVecEnv.__init__(self, len(specs), spec.observation_space, spec.share_observation_space, spec.action_space)
        self.node_observation_space = spec.node_observation_space
        self.adj_observation_space = spec.adj_observation_space
        self.edge_observation_space = spec.edge_observation_space
        self.agent_id_observation_space = spec.agent_id_observation_space
        self.share_agent_id_observation_space = spec.share_agent_id_observation_space
        self.actions = None
        self._staging = None
        self._act_pin = None

    def step_async(self, actions):
        # (n, N, 5) one-hot / continuous floats or (n, N) indices; through a pinned buffer (pageable copies stall)
        a = np.asarray(actions)
        a = a.astype(np.float32 if a.ndim == 3 else np.int32, copy=False)
        if self._act_pin is None or tuple(self._act_pin.shape) != a.shape or self._act_pin.numpy().dtype != a.dtype:
            self._act_pin = torch.empty(a.shape, dtype=torch.float32 if a.ndim == 3 else torch.int32, pin_memory=True)
        self._act_pin.numpy()[...] = a
        # safe to reuse next step: step_wait ends with a stream synchronisation
        self.actions = self._act_pin.to(self.engine.device, non_blocking=True)

    def _step_device(self):
        obs, ids, node, adj, rew, done, info = self.engine.step(self.actions, auto_reset=True)
        self.actions = None
        if self.spec.cfg.collaborative:
            # reference multiagent/environment.py:867-870: every agent gets [sum of the env's rewards]
            # (a list of 1-element lists -> shape (n, N, 1)); infos keep the individual rewards
            rew = rew.sum(dim=1, keepdim=True).expand_as(rew).unsqueeze(-1)
        return obs, ids, node, adj, rew, done, info

    def _fetch(self, *tensors):
        """Device tensors -> fresh float64 NumPy arrays of the same shapes (what the reference's workers pipe back).
        Everything crosses PCIe in ONE copy into a pinned staging buffer, as float32 -- what the device holds: half the bytes of the
        float64 the caller gets -- and is widened (exactly) by the copy that cuts the caller-owned arrays out of the staging buffer,
        a pass the host makes anyway (round 6; before, the widening happened on the device and twice the bytes crossed)."""
        dt = torch.float32 if os.environ.get('FMARL_FETCH_F64') != '1' else torch.float64   # (=1: the device-side widening of rounds 1-5, for A/B)
        flat = torch.cat([t.detach().to(dt).reshape(-1) for t in tensors])
        if self._staging is None or self._staging.numel() < flat.numel() or self._staging.dtype != dt:
            self._staging = torch.empty(flat.numel(), dtype=dt, pin_memory=True)
        host = self._staging[:flat.numel()]
        host.copy_(flat, non_blocking=True)
        torch.cuda.current_stream(self.engine.device).synchronize()
        arr, out, o, jobs = host.numpy(), [], 0, []
        for slot, t in enumerate(tensors):
            dst = self._fresh(tuple(t.shape), slot)
            flat_dst, m = dst.reshape(-1), t.numel()
            if m >= self.PARALLEL_COPY_ELEMS and self._copy_pool() is not None:
                # large outputs (node_obs at 512 envs x 32 agents: 104 MB of float64): the widening copy in slices on a few host threads
                # (NumPy releases the GIL inside copyto) -- one thread fills them at 8 GB/s, 13 ms of a 16 ms step
                k = self._copy_threads
                cuts = [o + m * j // k for j in range(k + 1)]
                jobs += [self._pool.submit(np.copyto, flat_dst[a - o:b - o], arr[a:b]) for a, b in zip(cuts, cuts[1:])]
            else:
                np.copyto(flat_dst, arr[o:o + m])
            out.append(dst)
            o += m
        for j in jobs:
            j.result()
        return out

    PARALLEL_COPY_ELEMS = 4 << 20   # outputs of four million elements and more are widened by several host threads (below, starting them costs more: 4 096 x 3 went 0.71 -> 0.9 ms)
    _pool, _copy_threads = None, 0

    def _copy_pool(self):
        if self._pool is None and self._copy_threads == 0:
            k = min(8, (os.cpu_count() or 1) // 2)
            self._copy_threads = k if k >= 2 else -1
            if k >= 2:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=k, thread_name_prefix='fmarl-copy')
        return self._pool

    def _fresh(self, shape, slot=0):
        """A float64 array for output number ``slot`` of this call: new (the caller's for good) unless ``reuse_outputs = k``
        opted into a ring of k generations per output."""
        k = int(self.reuse_outputs)
        if k <= 0:
            return np.empty(shape, dtype=np.float64)
        ring = self.__dict__.setdefault('_rings', {}).setdefault((slot, shape), [[], 0])
        if len(ring[0]) < k:
            ring[0].append(np.empty(shape, dtype=np.float64))
            return ring[0][-1]
        ring[1] = (ring[1] + 1) % k
        return ring[0][ring[1] - 1]

    # beyond this many (env, agent) dicts per step the infos stay a lazy view (65 536 x 32 dicts per step cannot be built)
    EAGER_INFOS = 256

    def _infos(self, records, as_array):
        """infos in the reference's own types while that is affordable: GraphSubprocVecEnv / SubprocVecEnv return a tuple
        over envs of lists of per-agent dicts (env_wrappers.py:992, :274), the Dummy variants ``np.array`` of the same
        (:916, :699: an object array (n, N)).  Large batches get ``LazyInfos``, which offers the access pattern the runner
        uses (base_runner.py:208-243: iterate envs, index the agent, ``.keys()`` / ``[key]``)."""
        lazy = LazyInfos(records, self.spec.cfg.scenario_name)
        n, N = records.shape[0], records.shape[1]
        if n * N > self.EAGER_INFOS:
            return lazy
        dicts = [[lazy[e][a] for a in range(N)] for e in range(n)]
        if not as_array:
            return tuple(dicts)
        arr = np.empty((n, N), dtype=object)
        for e in range(n):
            for a in range(N):
                arr[e, a] = dicts[e][a]
        return arr

    def _agent_ids(self):
        n, N = self.engine.n_envs, self.spec.cfg.N
        return np.broadcast_to(np.arange(N, dtype=np.int64).reshape(1, N, 1), (n, N, 1)).copy()

    def close_extras(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        self.engine.close()

    def reset_task(self):
        raise NotImplementedError  # reference worker: env.reset_task() does not exist for these scenarios


class GraphSubprocVecEnv(_EngineVecEnv):
    """reference env_wrappers.py:951-1026 (spaces argument accepted and ignored like the reference)."""
    _infos_as_array = False

    def __init__(self, env_fns, spaces=None, device='cuda:0'):
        _EngineVecEnv.__init__(self, env_fns, device)

    def _adj(self, adj_env):
        n, N, E = self.engine.n_envs, self.spec.cfg.N, self.spec.cfg.E
        view = np.broadcast_to(adj_env[:, None], (n, N, E, E))
        return view.copy() if self.materialize_adj else view

    def step_wait(self):
        obs, ids, node, adj, rew, done, info = self._step_device()
        obs, node, adj_env, rew, done, info = self._fetch(obs, node, self.engine.adj_env, rew, done, info)
        return obs, self._agent_ids(), node, self._adj(adj_env), rew, done != 0, self._infos(info, self._infos_as_array)

    def reset(self):
        obs, ids, node, adj = self.engine.reset()
        obs, node, adj_env = self._fetch(obs, node, self.engine.adj_env)
        return obs, self._agent_ids(), node, self._adj(adj_env)


class GraphDummyVecEnv(GraphSubprocVecEnv):
    """reference env_wrappers.py:895-948: same data plus ``reset_count`` as 8th item; infos as an object array."""
    _infos_as_array = True

    def __init__(self, env_fns, device='cuda:0'):
        _EngineVecEnv.__init__(self, env_fns, device)

    def step_wait(self):
        res = GraphSubprocVecEnv.step_wait(self)
        reset_count = 1 if bool(res[5].all(axis=1).any()) else 0
        return res + (reset_count,)


class SubprocVecEnv(_EngineVecEnv):
    """reference env_wrappers.py:242-307 (env_name == 'MPE': graph outputs dropped)."""
    _infos_as_array = False

    def __init__(self, env_fns, spaces=None, device='cuda:0'):
        _EngineVecEnv.__init__(self, env_fns, device, emit_graph=False)

    def step_wait(self):
        obs, ids, node, adj, rew, done, info = self._step_device()
        obs, rew, done, info = self._fetch(obs, rew, done, info)
        return obs, rew, done != 0, self._infos(info, self._infos_as_array)

    def reset(self):
        obs, ids, node, adj = self.engine.reset()
        return self._fetch(obs)[0]


class DummyVecEnv(SubprocVecEnv):
    """reference env_wrappers.py:686-729"""
    _infos_as_array = True

    def __init__(self, env_fns, device='cuda:0'):
        _EngineVecEnv.__init__(self, env_fns, device, emit_graph=False)
