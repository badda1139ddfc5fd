"""Device-resident rollout buffer in the layout of the reference's ``GraphReplayBuffer``
(reference onpolicy/utils/graph_buffer.py:84-165: ``(T + 1, n_rollout_threads, num_agents, ...)``
float32 arrays), filled **in place** by the step kernels.

The reference copies every step's NumPy outputs into the buffer on the host
(onpolicy/runner/shared/graph_mpe_runner.py:438-488 ``insert``).  Here each time slot is an output
set of the engine, so "insert" is free: step t writes obs / node_obs / adj straight into slot t + 1 and
rewards into slot t.  What ``insert`` computes on the host is done with a few device ops:

* ``share_obs`` (centralized critic, graph_mpe_runner.py:470-478): every agent sees the concatenation of all
  agents' obs -> a stride-0 view ``obs.reshape(T+1, n, 1, N*D).expand(.., N, ..)``, never materialised;
* ``adj`` is stored once per env and exposed per agent as a stride-0 view (the reference stores N copies);
* ``masks`` = 0 where the agent was done, ``active_masks`` = 0 where an agent is done but its env is not
  (graph_mpe_runner.py:452-465).

At BASELINE config 3 (65 536 envs x 32 agents, E = 72) one slot is 8.3 GB, a 25-step episode 216 GB:
it fits the 288 GB of one MI355X, which is the point of keeping it resident.

The learner's side: ``attach_policy`` adds the slots the policy fills (value_preds, actions,
action_log_probs, rnn states, bad_masks, available_actions: graph_buffer.py:116-164), ``compute_returns`` is
``GraphReplayBuffer.compute_returns`` (:285-366, every branch, bit for bit), ``advantages`` what ``GR_MAPPO.train`` forms
before its epochs (onpolicy/algorithms/graph_mappo.py:294-304), ``feed_forward_generator`` / ``recurrent_generator`` the
reference's minibatch generators (:368-453, :597-758) yielding the same 16-tuples as device tensors -- one gather kernel per
minibatch (``fmarl_minibatch_gather``) instead of NumPy fancy indexing on the host.
"""
import ctypes as C

import torch

from . import _lib


class DeviceRolloutBuffer(object):
    def __init__(self, engine, episode_length=None):
        self.engine = eng = engine
        cfg = eng.cfg
        self.T = T = int(episode_length or cfg.episode_length)
        n, N, E, D, F = eng.n_envs, cfg.N, cfg.E, cfg.obs_dim, cfg.node_feat
        dev = eng.device
        z = lambda *shape, dtype=torch.float32: torch.zeros(*shape, dtype=dtype, device=dev)  # noqa: E731
        self.obs = z(T + 1, n, N, D)
        # (the two large arrays: slots virtually contiguous, physical memory interleaved over the whole array -- a step that fills ONE
        # slot then writes at the rate of the whole buffer: engine.alloc_time_slots / fmarl_ring_alloc)
        from .engine import alloc_time_slots
        self.node_obs, _ = alloc_time_slots(eng.lib, dev, (T + 1, n, N, E, F), zero=True)
        self.adj_env, _ = alloc_time_slots(eng.lib, dev, (T + 1, n, E, E), zero=True)
        self.rewards = z(T, n, N, 1)
        self.dones = z(T, n, N, dtype=torch.uint8)
        self.masks = torch.ones(T + 1, n, N, 1, dtype=torch.float32, device=dev)
        self.active_masks = torch.ones(T + 1, n, N, 1, dtype=torch.float32, device=dev)
        self.agent_id = torch.arange(N, dtype=torch.int32, device=dev).view(1, 1, N, 1).expand(T + 1, n, N, 1)
        self.share_agent_id = torch.arange(N, dtype=torch.int32, device=dev).view(1, 1, 1, N).expand(T + 1, n, N, N)
        self.info_planes = z(T + 1, 14, n, N) if eng.emit_info else None   # one set of info planes per slot (the last step's: process_infos)
        self.value_preds = self.returns = self.bad_masks = self.actions = self.action_log_probs = None   # attach_policy()
        self.rnn_states = self.rnn_states_critic = self.available_actions = None
        self._scratch_reward = z(n, N)
        self._scratch_done = z(n, N, dtype=torch.uint8)
        # slot t receives the observation that FOLLOWS step t - 1; reward / done of step t go to index t
        self._sets = []
        for t in range(T + 1):
            rew = self.rewards[t - 1].view(n, N) if t >= 1 else self._scratch_reward
            done = self.dones[t - 1] if t >= 1 else self._scratch_done
            self._sets.append(eng.new_output_set(obs=self.obs[t], reward=rew, done=done, node_obs=self.node_obs[t], adj_env=self.adj_env[t],
                                                 info_planes=self.info_planes[t] if self.info_planes is not None else None))
        self.step = 0

    # views in the reference's shapes ---------------------------------------------------------------
    @property
    def adj(self):
        T1, n, E = self.adj_env.shape[0], self.adj_env.shape[1], self.adj_env.shape[2]
        return self.adj_env.view(T1, n, 1, E, E).expand(T1, n, self.engine.cfg.N, E, E)

    @property
    def share_obs(self):
        T1, n, N, D = self.obs.shape
        return self.obs.view(T1, n, 1, N * D).expand(T1, n, N, N * D)

    # rollout -----------------------------------------------------------------------------------------
    def reset(self):
        """envs.reset() -> slot 0 (reference graph_mpe_runner.py:178-203 warmup)."""
        self.engine.use_outputs(self._sets[0])
        self.engine.reset()
        self.step = 0
        self.masks.fill_(1.0)
        self.active_masks.fill_(1.0)

    def insert_step(self, actions):
        """envs.step(actions) + buffer.insert of the env outputs: slot step + 1, rewards[step]."""
        t = self.step
        if t >= self.T:
            raise RuntimeError('buffer full: call after_update() first')
        self.engine.use_outputs(self._sets[t + 1])
        self.engine.step(actions, auto_reset=True)
        self._insert_masks(t, 1)   # graph_mpe_runner.py:444-465
        self.step = t + 1
        return self._sets[t + 1]

    def _insert_masks(self, t0, T):
        """masks / active_masks of the slots t0 + 1 ... t0 + T from the dones of the steps t0 ... t0 + T - 1 (one launch)."""
        eng = self.engine
        n, N = eng.n_envs, eng.cfg.N
        with torch.cuda.device(eng.device):
            _lib.check(eng.lib.fmarl_insert_masks(self.dones[t0].data_ptr(), self.masks[t0 + 1].data_ptr(), self.active_masks[t0 + 1].data_ptr(),
                                                  T * n, N, eng._stream()), 'fmarl_insert_masks')

    def insert_span(self, action_tape):
        """``insert_step(action_tape[t])`` for every t through ONE ``fmarl_step_span`` call: the time slots are contiguous
        (T + 1, n, ...) arrays, so step t's obs / node_obs / adj land in slot ``step + t + 1`` and its reward / done at index
        ``step + t`` by per-step strides; the masks of the runner's insert are formed for all steps at once afterwards.
        Same buffer contents as T ``insert_step`` calls, bit for bit, the per-slot info planes included."""
        eng, t0 = self.engine, self.step
        tape = action_tape.to(eng.device)
        T = int(tape.shape[0])
        if t0 + T > self.T:
            raise RuntimeError('tape of %d steps does not fit behind step %d of %d' % (T, t0, self.T))
        n, N = eng.n_envs, eng.cfg.N
        first = self._sets[t0 + 1]
        eng.use_outputs(first)
        eng.step_span(tape, strides=dict(obs=self.obs[0].numel(), node_obs=self.node_obs[0].numel(), adj=self.adj_env[0].numel(),
                                         reward=n * N, done=n * N, info=self.info_planes[0].numel() if self.info_planes is not None else 0))
        self._insert_masks(t0, T)
        self.step = t0 + T
        last = self._sets[self.step]
        eng.use_outputs(last)
        return last

    def capture(self, action_tape):
        """Capture ``insert_step(action_tape[t])`` for every t -- the step kernels writing their slots AND the masks /
        active_masks of the runner's insert -- into one hipGraph.  ``replay()`` on the returned object runs the rollout
        with one launch and leaves ``step`` where eager inserts would.  ``action_tape`` (T', n, N) int32 or (T', n, N, 5)
        float32 is read at replay time; T' <= T - step.  Needs an engine with ``async_reset=False``."""
        eng, dev = self.engine, self.engine.device
        tape = action_tape.to(dev)
        start = self.step
        if start + tape.shape[0] > self.T:
            raise RuntimeError('tape of %d steps does not fit behind step %d of %d' % (tape.shape[0], start, self.T))
        import gc
        gc.collect()   # nothing may be destroyed (streams, events, other graphs) while this stream captures
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.device(dev), torch.cuda.graph(graph):
            for t in range(tape.shape[0]):
                self.insert_step(tape[t])
        self.step = start
        return _CapturedInserts(self, graph, tape, start)

    def after_update(self):
        """graph_buffer.py after_update: the last slot becomes the first of the next rollout."""
        names = ['obs', 'node_obs', 'adj_env', 'masks', 'active_masks']
        if self.value_preds is not None:
            names += ['rnn_states', 'rnn_states_critic', 'bad_masks', 'available_actions']
        for name in names:
            buf = getattr(self, name)
            if buf is not None:
                buf[0].copy_(buf[-1])
        self.step = 0

    # the learner's side ------------------------------------------------------------------------------
    def attach_policy(self, act_dim=1, recurrent_N=1, hidden_size=64, n_actions=5):
        """The slots the policy fills, in the reference's shapes (graph_buffer.py:116-164): value_preds / returns / bad_masks
        (T + 1, n, N, 1), actions / action_log_probs (T, n, N, act_dim), rnn_states / rnn_states_critic
        (T + 1, n, N, recurrent_N, hidden_size), available_actions (T + 1, n, N, n_actions) of ones (None if n_actions is 0)."""
        T, n, N, dev = self.T, self.engine.n_envs, self.engine.cfg.N, self.engine.device
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        self.value_preds, self.returns = z(T + 1, n, N, 1), z(T + 1, n, N, 1)
        self.bad_masks = torch.ones(T + 1, n, N, 1, dtype=torch.float32, device=dev)
        self.actions, self.action_log_probs = z(T, n, N, act_dim), z(T, n, N, act_dim)
        self.rnn_states, self.rnn_states_critic = z(T + 1, n, N, recurrent_N, hidden_size), z(T + 1, n, N, recurrent_N, hidden_size)
        self.available_actions = torch.ones(T + 1, n, N, n_actions, dtype=torch.float32, device=dev) if n_actions else None
        self._adv_ws = torch.zeros(_lib.load().fmarl_advantages_workspace(), dtype=torch.uint8, device=dev)
        return self

    def insert_policy(self, t, value_preds, actions, action_log_probs, rnn_states=None, rnn_states_critic=None, bad_masks=None,
                      available_actions=None):
        """The policy's part of ``GraphReplayBuffer.insert`` for buffer step t (graph_buffer.py:233-248): values, actions and
        log-probabilities at index t, the recurrent states / bad_masks / available_actions that follow the step at t + 1."""
        self._need_policy()
        self.value_preds[t].copy_(value_preds.reshape(self.value_preds[t].shape))
        self.actions[t].copy_(actions.reshape(self.actions[t].shape))
        self.action_log_probs[t].copy_(action_log_probs.reshape(self.action_log_probs[t].shape))
        for dst, src in ((self.rnn_states, rnn_states), (self.rnn_states_critic, rnn_states_critic), (self.bad_masks, bad_masks),
                         (self.available_actions, available_actions)):
            if src is not None:
                dst[t + 1].copy_(src.reshape(dst[t + 1].shape))

    def _need_policy(self):
        if self.value_preds is None:
            raise RuntimeError('call attach_policy() first: the buffer holds only the env-facing arrays')

    @staticmethod
    def _mean_std(value_normalizer):
        """(denormalize?, mean, stddev) of a value normaliser: None, a (mean, stddev) pair, or the reference's ValueNorm /
        PopArt objects (``running_mean_var`` / ``debiased_mean_var``: onpolicy/utils/valuenorm.py:47-54,
        onpolicy/algorithms/utils/popart.py:85-89) -- stddev = float32 sqrt of the debiased variance, as ``denormalize`` forms it."""
        if value_normalizer is None:
            return 0, 0.0, 1.0
        if isinstance(value_normalizer, (tuple, list)):
            return 1, float(value_normalizer[0]), float(value_normalizer[1])
        fn = getattr(value_normalizer, 'running_mean_var', None) or getattr(value_normalizer, 'debiased_mean_var')
        mean, var = fn()
        return 1, float(mean.reshape(-1)[0]), float(torch.sqrt(var.to(torch.float32)).reshape(-1)[0])

    def compute_returns(self, next_value, value_normalizer=None, gamma=0.99, gae_lambda=0.95, use_gae=True, use_proper_time_limits=False):
        """``GraphReplayBuffer.compute_returns`` (graph_buffer.py:285-366) on the device, all branches: fills ``returns`` (and
        ``value_preds[-1]`` with GAE) from rewards / value_preds / masks / bad_masks.  ``next_value`` (n, N, 1) float32;
        ``value_normalizer``: see ``_mean_std``.  Equal to the reference's NumPy result bit for bit."""
        self._need_policy()
        dn, mean, std = self._mean_std(value_normalizer)
        n, N = self.engine.n_envs, self.engine.cfg.N
        nv = next_value.to(device=self.engine.device, dtype=torch.float32).reshape(n * N).contiguous()
        args = _lib.FmarlReturns(float(gamma), float(gae_lambda), mean, std, dn, int(bool(use_gae)), int(bool(use_proper_time_limits)),
                                 self.T, n * N)
        rc = _lib.load().fmarl_compute_returns(C.byref(args), self.rewards.data_ptr(), self.value_preds.data_ptr(), self.masks.data_ptr(),
                                               self.bad_masks.data_ptr(), nv.data_ptr(), self.returns.data_ptr(),
                                               torch.cuda.current_stream(self.engine.device).cuda_stream)
        _lib.check(rc, 'fmarl_compute_returns')
        self._keep = nv
        return self.returns

    def advantages(self, value_normalizer=None, group=None):
        """What ``GR_MAPPO.train`` hands to the generators (onpolicy/algorithms/graph_mappo.py:294-304): returns[:-1] minus the
        (denormalised) value predictions, standardised over the entries with a non-zero active mask.  (T, n, N, 1) float32;
        ``advantage_stats`` afterwards views the (mean, std) the kernel used.

        ``group``: a ``torch.distributed`` process group (or True for the default one) of data-parallel learners, one rollout
        shard and one buffer per GPU: the (count, sum, sum of squares) triples of the ranks are added (one 24-byte all-reduce)
        and every rank standardises with the statistics of the whole batch, as one process holding all envs would."""
        self._need_policy()
        dn, mean, std = self._mean_std(value_normalizer)
        adv = torch.empty_like(self.rewards)
        lib, st = _lib.load(), torch.cuda.current_stream(self.engine.device).cuda_stream
        args = (self.returns.data_ptr(), self.value_preds.data_ptr(), self.active_masks.data_ptr(), adv.data_ptr(), adv.numel(), dn, mean, std,
                self._adv_ws.data_ptr(), st)
        if group is None:
            _lib.check(lib.fmarl_advantages(*args), 'fmarl_advantages')
        else:
            import torch.distributed as dist
            grp = None if group is True else group
            _lib.check(lib.fmarl_advantages_sums(*args), 'fmarl_advantages_sums')
            sums = self._adv_ws[16:40].view(torch.float64)
            if dist.get_backend(grp) == 'gloo':      # (ranks sharing a GPU in the tests: through host memory)
                host = sums.cpu()
                dist.all_reduce(host, group=grp)
                sums.copy_(host)
            else:
                dist.all_reduce(sums, group=grp)     # RCCL, on the current stream's successor: 24 bytes
            _lib.check(lib.fmarl_advantages_apply(adv.data_ptr(), adv.numel(), self._adv_ws.data_ptr(), st), 'fmarl_advantages_apply')
        self.advantage_stats = self._adv_ws[:8].view(torch.float32)
        return adv

    def _gather(self, adv, index, rows, mode, chunk, want):
        eng, cfg = self.engine, self.engine.cfg
        dev, n, N = eng.device, eng.n_envs, cfg.N
        R = rows if mode == 0 else rows // chunk   # rows of the two rnn-state outputs
        f = lambda *shape, dtype=torch.float32: torch.empty(*shape, dtype=dtype, device=dev)  # noqa: E731
        shapes = dict(share_obs=(rows, N * cfg.obs_dim), obs=(rows, cfg.obs_dim), node_obs=(rows, cfg.E, cfg.node_feat), adj=(rows, cfg.E, cfg.E),
                      agent_id=(rows, 1), share_agent_id=(rows, N), rnn_states=(R,) + tuple(self.rnn_states.shape[3:]),
                      rnn_states_critic=(R,) + tuple(self.rnn_states_critic.shape[3:]), actions=(rows, self.actions.shape[-1]),
                      value_preds=(rows, 1), returns=(rows, 1), masks=(rows, 1), active_masks=(rows, 1),
                      old_action_log_probs=(rows, self.action_log_probs.shape[-1]), adv_targ=(rows, 1),
                      available_actions=(rows, self.available_actions.shape[-1]) if self.available_actions is not None else None,
                      env_slot=(rows,))
        out = {}
        for k in _lib.BATCH_DST_ARRAYS:
            if k not in want or shapes[k] is None or (k == 'adv_targ' and adv is None):
                out[k] = None
            else:
                out[k] = f(*shapes[k], dtype=torch.int32 if k in ('agent_id', 'share_agent_id') else torch.int64 if k == 'env_slot' else torch.float32)
        ptr = lambda x: x.data_ptr() if x is not None else None  # noqa: E731
        src = _lib.FmarlBatchSrc(ptr(self.obs), ptr(self.node_obs), ptr(self.adj_env), ptr(self.rnn_states), ptr(self.rnn_states_critic),
                                 ptr(self.actions), ptr(self.action_log_probs), ptr(self.value_preds), ptr(self.returns), ptr(self.masks),
                                 ptr(self.active_masks), ptr(adv), ptr(self.available_actions), self.T, n, N, cfg.obs_dim, cfg.E, cfg.node_feat,
                                 int(self.rnn_states[0, 0, 0].numel()), int(self.actions.shape[-1]),
                                 int(self.available_actions.shape[-1]) if self.available_actions is not None else 0, 0)
        dst = _lib.FmarlBatchDst(*[ptr(out[k]) for k in _lib.BATCH_DST_ARRAYS])
        rc = _lib.load().fmarl_minibatch_gather(C.byref(src), C.byref(dst), index.data_ptr(), rows, mode, chunk,
                                                torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, 'fmarl_minibatch_gather')
        return out

    GENERATOR_FIELDS = _lib.BATCH_DST_ARRAYS[:16]

    def _generate(self, adv, perm, size, batches, mode, chunk, fields, with_env_slot):
        self._need_policy()
        dev = self.engine.device
        if adv is not None:
            adv = adv.to(device=dev, dtype=torch.float32).contiguous()
            if adv.numel() != self.rewards.numel():
                raise ValueError('advantages must have %d entries (T, n, N, 1)' % self.rewards.numel())
        want = set(fields or self.GENERATOR_FIELDS) | ({'env_slot'} if with_env_slot else set())
        perm = perm.to(device=dev, dtype=torch.int64).contiguous()
        for b in range(batches):
            index = perm[b * size:(b + 1) * size]
            out = self._gather(adv, index, size * (chunk if mode else 1), mode, chunk, want)
            tup = tuple(out[k] for k in self.GENERATOR_FIELDS)
            yield tup + (out['env_slot'],) if with_env_slot else tup

    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None, perm=None, fields=None, with_env_slot=False):
        """``GraphReplayBuffer.feed_forward_generator`` (graph_buffer.py:368-453): yields, per minibatch, the reference's
        16-tuple (share_obs, obs, node_obs, adj, agent_id, share_agent_id, rnn_states, rnn_states_critic, actions, value_preds,
        returns, masks, active_masks, old_action_log_probs, adv_targ, available_actions) as device tensors gathered by one
        kernel.  ``perm``: the permutation of range(T n N) to use (default: ``torch.randperm`` on the host generator, like the
        reference).  ``fields``: the subset to materialise (the others come back as None) -- at BASELINE scale a policy takes
        ``with_env_slot=True`` and indexes ``adj_env.view(-1, E, E)`` instead of receiving N copies of every matrix."""
        n, N = self.engine.n_envs, self.engine.cfg.N
        batch = self.T * n * N
        if mini_batch_size is None:
            if batch < num_mini_batch:
                raise ValueError('PPO requires processes (%d) * steps (%d) * agents (%d) >= mini batches (%d)' % (n, self.T, N, num_mini_batch))
            mini_batch_size = batch // num_mini_batch
        else:
            num_mini_batch = num_mini_batch or batch // mini_batch_size
        if perm is None:
            perm = torch.randperm(batch)
        return self._generate(advantages, perm, mini_batch_size, num_mini_batch, 0, 1, fields, with_env_slot)

    def recurrent_generator(self, advantages, num_mini_batch, data_chunk_length, perm=None, fields=None, with_env_slot=False):
        """``GraphReplayBuffer.recurrent_generator`` (graph_buffer.py:597-758): chunks of ``data_chunk_length`` consecutive
        entries of the (n, N, T)-ordered series, ``data_chunks // num_mini_batch`` permuted chunks per minibatch, rows ordered
        (L, chunk) like the reference's stack + flatten; rnn states one row per chunk.  ``perm`` permutes range(data_chunks)."""
        n, N = self.engine.n_envs, self.engine.cfg.N
        chunks = self.T * n * N // data_chunk_length
        if perm is None:
            perm = torch.randperm(chunks)
        return self._generate(advantages, perm, chunks // num_mini_batch, num_mini_batch, 1, int(data_chunk_length), fields, with_env_slot)


class _CapturedInserts(object):
    """A captured run of ``insert_step`` calls (DeviceRolloutBuffer.capture)."""

    def __init__(self, buf, graph, tape, start):
        self.buf, self.graph, self.tape, self.start = buf, graph, tape, start

    def replay(self):
        if self.buf.step != self.start:
            raise RuntimeError('captured inserts start at buffer step %d, the buffer is at %d' % (self.start, self.buf.step))
        self.graph.replay()
        self.buf.step = self.start + self.tape.shape[0]
        self.buf.engine.use_outputs(self.buf._sets[self.buf.step])
