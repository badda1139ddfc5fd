"""Sharding of the rollout over the GPUs of one node and the per-step trajectory exchange.

The reference has exactly one kind of parallelism: ``n_rollout_threads`` independent env
processes (reference onpolicy/envs/env_wrappers.py:951-1026).  Environments never interact, so
here each rank (one process per GPU) owns a contiguous range of env indices and runs them with
its own ``RolloutEngine``; the Philox streams are keyed by the GLOBAL env index
(``FmarlConfig.env_offset``), so results do not depend on the number of GPUs.

The only exchange step is the one the reference does through its worker pipes
(env_wrappers.py:988-996): handing the per-step trajectory record to the learner.  Here that is
one ``gather`` per step to the learner rank over RCCL (backend "nccl" on ROCm; xGMI links are
point to point, so each peer's shard moves over its own link).  The record is the compact part
of the step output -- obs (which already carries every agent's velocity, position and goal
offset), reward and done: 33 bytes per agent-step.  node_obs / adj are not sent: for navigation_graph
the learner rebuilds them (``RolloutEngine.rebuild_graph``) from the obs rows plus the episode record
(goals, landmarks, obstacles, walls: ``RolloutEngine.pack_episode``), which is gathered once per
episode because World.step never moves those entities.  Records are double-buffered: the gather of
step t runs on RCCL's stream while the kernels of step t+1 write the other buffer.
"""
import torch
import torch.distributed as dist


def shard_range(n_total, world_size, rank):
    """Contiguous env range [lo, hi) of ``rank``; the first ``n_total % world_size`` ranks get one more."""
    base, rem = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class StepRecord(object):
    """One flat byte buffer laid out as [obs f32 (n,N,D) | reward f32 (n,N) | graph record i32 (n,N,G) | done u8 (n,N) | pad].
    ``graph_words`` = RolloutEngine.step_record_words (0 for navigation_graph: its obs rows are the record)."""

    def __init__(self, n_envs, num_agents, obs_dim, device, graph_words=0):
        self.shape = (int(n_envs), int(num_agents), int(obs_dim))
        self.graph_words = int(graph_words)
        n, N, D = self.shape
        self.obs_bytes, self.rew_bytes, self.done_bytes = n * N * D * 4, n * N * 4, n * N
        self.graph_bytes = n * N * self.graph_words * 4
        total = (self.obs_bytes + self.rew_bytes + self.graph_bytes + self.done_bytes + 15) // 16 * 16
        self.flat = torch.zeros(total, dtype=torch.uint8, device=device)
        self.obs, self.reward, self.done = self.views(self.flat)
        self.graph = self.graph_view(self.flat)

    def views(self, flat):
        n, N, D = self.shape
        a, b = self.obs_bytes, self.obs_bytes + self.rew_bytes
        c = b + self.graph_bytes
        return (flat[:a].view(torch.float32).view(n, N, D), flat[a:b].view(torch.float32).view(n, N),
                flat[c:c + self.done_bytes].view(n, N))

    def graph_view(self, flat):
        """(n, N, graph_words) int32 view of the per-step graph record (None when the scenario has none)."""
        if not self.graph_words:
            return None
        n, N, _ = self.shape
        b = self.obs_bytes + self.rew_bytes
        return flat[b:b + self.graph_bytes].view(torch.int32).view(n, N, self.graph_words)

    @staticmethod
    def bytes_per_agent_step(obs_dim, graph_words=0):
        return obs_dim * 4 + 4 + 1 + 4 * graph_words


class SpanRecord(object):
    """T step records back to back in one flat byte buffer -- record t = the StepRecord layout at byte ``t * stride`` -- so that
    a run of steps (``RolloutEngine.step_span``) writes them through per-step strides and ONE collective moves the whole run."""

    def __init__(self, T, n_envs, num_agents, obs_dim, device, graph_words=0):
        self.T, self.shape, self.graph_words = int(T), (int(n_envs), int(num_agents), int(obs_dim)), int(graph_words)
        n, N, D = self.shape
        G = self.graph_words
        self.obs_bytes, self.rew_bytes, self.graph_bytes, self.done_bytes = n * N * D * 4, n * N * 4, n * N * G * 4, n * N
        self.stride = (self.obs_bytes + self.rew_bytes + self.graph_bytes + self.done_bytes + 15) // 16 * 16
        self.flat = torch.zeros(self.T * self.stride, dtype=torch.uint8, device=device)
        self.obs, self.reward, self.done = self.views(self.flat, self.T)
        self.graph = self.graph_view(self.flat, self.T)
        # per-step element strides for fmarl_step_span (float / int32 arrays: stride / 4 elements, done: bytes)
        self.strides = dict(obs=self.stride // 4, reward=self.stride // 4, done=self.stride, graph_record=self.stride // 4)

    def views(self, flat, steps):
        """(obs (steps, n, N, D), reward (steps, n, N), done (steps, n, N)) strided views of the first ``steps`` records."""
        n, N, D = self.shape
        f32, w = flat.view(torch.float32), self.stride // 4
        obs = f32.as_strided((steps, n, N, D), (w, N * D, D, 1), 0)
        rew = f32.as_strided((steps, n, N), (w, N, 1), self.obs_bytes // 4)
        done = flat.as_strided((steps, n, N), (self.stride, N, 1), self.obs_bytes + self.rew_bytes + self.graph_bytes)
        return obs, rew, done

    def graph_view(self, flat, steps):
        if not self.graph_words:
            return None
        n, N, _ = self.shape
        G = self.graph_words
        return flat.view(torch.int32).as_strided((steps, n, N, G), (self.stride // 4, N * G, G, 1), (self.obs_bytes + self.rew_bytes) // 4)

    def first_step_buffers(self):
        """Contiguous tensors of step 0 (what an OutputSet is built from; the span strides reach the other steps)."""
        n, N, D = self.shape
        a, b = self.obs_bytes, self.obs_bytes + self.rew_bytes
        c = b + self.graph_bytes
        obs = self.flat[:a].view(torch.float32).view(n, N, D)
        rew = self.flat[a:b].view(torch.float32).view(n, N)
        graph = self.flat[b:c].view(torch.int32).view(n, N, self.graph_words) if self.graph_words else None
        done = self.flat[c:c + self.done_bytes].view(n, N)
        return obs, rew, done, graph


class TrajectoryGather(object):
    """Double-buffered asynchronous gather of StepRecords to the learner rank.

    Usage per step t:  rec = tg.record(t)  ->  engine writes rec.obs / rec.reward / rec.done  ->
    tg.submit(t).  ``record(t)`` first waits for the gather that last used the same buffer
    (step t - depth).  ``finish()`` waits for everything in flight.
    All ranks must use the same n_envs per rank (equal shards) -- gather needs equal sizes.
    """

    def __init__(self, n_envs, num_agents, obs_dim, device, group=None, dst=0, depth=2, episode_words=0,
                 force_collective=False, graph_words=0, timing=False):
        self._setup_exchange(n_envs, device, group, dst, depth, episode_words, force_collective, timing)
        self.records = [StepRecord(n_envs, num_agents, obs_dim, device, graph_words) for _ in range(depth)]
        self.pending = [None] * depth
        self.recv = None
        if self.collective and self.rank == dst:
            self.recv = [[torch.zeros_like(r.flat) for _ in range(self.world)] for r in self.records]

    def _setup_exchange(self, n_envs, device, group, dst, depth, episode_words, force_collective, timing):
        """What the per-step and the per-run exchange share: the group, the wait clocks, the episode records."""
        self.group, self.dst, self.depth = group, dst, depth
        # timing: how long the rollout is held up by the exchange.  Two clocks, because an RCCL work's wait() only makes the
        # current STREAM wait (the host returns at once) while a gloo work's wait() blocks the HOST: ``host_wait_s``
        # accumulates the host time inside wait(), ``stream_wait_ms()`` the time the compute stream sat between an event
        # recorded before the wait and one recorded after it (read it after a device synchronisation).
        self.timing = bool(timing) and torch.device(device).type == 'cuda'
        self.host_wait_s, self._wait_events, self.waits = 0.0, [], 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # force_collective: issue the gathers even in a group of ONE rank (the RCCL path on a one-GPU box)
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        # episode records (int32 words per env), two in rotation: the one of the running episode stays readable
        # on the learner while the next one is gathered
        self.ep_send = [torch.zeros(int(n_envs), int(episode_words), dtype=torch.int32, device=device) for _ in range(2)]
        self.ep_recv, self.ep_pending, self.ep_count = None, [None, None], 0
        if self.collective and self.rank == dst:
            self.ep_recv = [[torch.zeros_like(b) for _ in range(self.world)] for b in self.ep_send]

    def _wait(self, work):
        import time
        self.waits += 1
        if self.timing:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        t0 = time.perf_counter()
        work.wait()
        self.host_wait_s += time.perf_counter() - t0
        if self.timing:
            b.record()
            self._wait_events.append((a, b))

    def stream_wait_ms(self):
        """Total time [ms] the compute stream spent inside the waits so far (synchronise the device first); restarts."""
        total = sum(a.elapsed_time(b) for a, b in self._wait_events)
        self._wait_events = []
        return total

    def record(self, t):
        k = t % self.depth
        if self.pending[k] is not None:
            self._wait(self.pending[k])
            self.pending[k] = None
        return self.records[k]

    def submit(self, t):
        """Start the gather of step t.  It overwrites the learner's receive buffer of step t - depth: read that step
        (``gathered(t - depth)``) before submitting this one."""
        if not self.collective:
            return
        k = t % self.depth
        self.pending[k] = dist.gather(self.records[k].flat, self.recv[k] if self.rank == self.dst else None,
                                      dst=self.dst, group=self.group, async_op=True)

    def finish(self):
        for k in range(self.depth):
            if self.pending[k] is not None:
                self._wait(self.pending[k])
                self.pending[k] = None
        for k in range(2):
            if self.ep_pending[k] is not None:
                self._wait(self.ep_pending[k])
                self.ep_pending[k] = None

    # -- once per episode (every rank at the same steps: ``RolloutEngine.episode_started`` after reset / step)
    def episode_record(self):
        """Buffer for the record of the episode that starts now: fill it (``engine.pack_episode(out=...)``), then
        ``submit_episode()``."""
        k = self.ep_count % 2
        if self.ep_pending[k] is not None:
            self._wait(self.ep_pending[k])
            self.ep_pending[k] = None
        return self.ep_send[k]

    def submit_episode(self):
        k = self.ep_count % 2
        self.ep_count += 1
        if self.collective:
            self.ep_pending[k] = dist.gather(self.ep_send[k], self.ep_recv[k] if self.rank == self.dst else None,
                                             dst=self.dst, group=self.group, async_op=True)

    def gathered_episode(self, back=0):
        """On the learner rank: list over ranks of the most recently submitted episode record (n_envs, words); ``back=1``:
        the one before it (a step gathered just before an episode ended still belongs to that episode)."""
        k = (self.ep_count - 1 - back) % 2
        if self.ep_pending[k] is not None:
            self.ep_pending[k].wait()
            self.ep_pending[k] = None
        return [self.ep_send[k]] if not self.collective else list(self.ep_recv[k])

    def gathered(self, t):
        """On the learner rank: list over ranks of (obs, reward, done) views of step t.  Waits for the gather of that
        step (the handle stays in place: ``record(t + depth)`` waits on it again, a no-op), so the views are complete
        when this returns; they are overwritten by ``submit(t + depth)``."""
        k = t % self.depth
        if self.pending[k] is not None:
            self.pending[k].wait()
        if not self.collective:
            r = self.records[k]
            return [(r.obs, r.reward, r.done)]
        return [self.records[k].views(f) for f in self.recv[k]]

    def gathered_graph(self, t):
        """On the learner rank: list over ranks of the per-step graph records (n, N, graph_words) int32 of step t."""
        k = t % self.depth
        if self.pending[k] is not None:
            self.pending[k].wait()
        if not self.collective:
            return [self.records[k].graph]
        return [self.records[k].graph_view(f) for f in self.recv[k]]


class SpanGather(TrajectoryGather):
    """The exchange for rollouts that run as spans (``RolloutEngine.step_span``): the records of a whole run of steps -- up to
    ``max_steps``, e.g. an episode -- are written back to back by the span launch and gathered to the learner rank with ONE
    collective per run (SURVEY section 8 e: "batch T steps per collective"), double-buffered like the per-step gather: the
    gather of run c moves over xGMI while run c + 1 is computed.  Episode records work as in ``TrajectoryGather``.

    Usage per run c of k steps:  rec = sg.span_record(c)  ->  engine.use_outputs(sg.output_set(engine, c));
    engine.step_span(tape, strides=rec.strides)  ->  sg.submit_span(c, k)."""

    def __init__(self, max_steps, n_envs, num_agents, obs_dim, device, group=None, dst=0, depth=2, episode_words=0,
                 force_collective=False, graph_words=0, timing=False):
        self._setup_exchange(n_envs, device, group, dst, depth, episode_words, force_collective, timing)
        self.records, self.pending, self.recv = [], [None] * depth, None   # (no per-step records: ``record`` / ``submit`` are not for spans)
        self.spans = [SpanRecord(max_steps, n_envs, num_agents, obs_dim, device, graph_words) for _ in range(depth)]
        self.span_pending, self.span_steps = [None] * depth, [0] * depth
        self.span_recv = None
        if self.collective and self.rank == dst:
            self.span_recv = [[torch.zeros_like(r.flat) for _ in range(self.world)] for r in self.spans]
        self._sets = {}

    def span_record(self, c):
        k = c % self.depth
        if self.span_pending[k] is not None:
            self._wait(self.span_pending[k])
            self.span_pending[k] = None
        return self.spans[k]

    def output_set(self, engine, c):
        """The engine's output set whose obs / reward / done / graph_record are step 0 of run c's record."""
        k = c % self.depth
        if (id(engine), k) not in self._sets:
            obs, rew, done, graph = self.spans[k].first_step_buffers()
            self._sets[(id(engine), k)] = engine.new_output_set(obs=obs, reward=rew, done=done, graph_record=graph if engine.emit_graph_record else None)
        return self._sets[(id(engine), k)]

    def submit_span(self, c, steps):
        """Start the gather of run c (its first ``steps`` records).  Overwrites the learner's receive buffer of run c - depth."""
        k = c % self.depth
        self.span_steps[k] = int(steps)
        if not self.collective:
            return
        nbytes = int(steps) * self.spans[k].stride
        recv = [f[:nbytes] for f in self.span_recv[k]] if self.rank == self.dst else None
        self.span_pending[k] = dist.gather(self.spans[k].flat[:nbytes], recv, dst=self.dst, group=self.group, async_op=True)

    def gathered_span(self, c):
        """On the learner rank: list over ranks of (obs (steps, n, N, D), reward, done, graph or None) of run c; waits for its
        gather (the views are overwritten by ``submit_span(c + depth)``)."""
        k = c % self.depth
        if self.span_pending[k] is not None:
            self.span_pending[k].wait()
        rec, steps = self.spans[k], self.span_steps[k]
        flats = [rec.flat] if not self.collective else self.span_recv[k]
        return [rec.views(f, steps) + (rec.graph_view(f, steps),) for f in flats]

    def finish(self):
        TrajectoryGather.finish(self)
        for k in range(self.depth):
            if self.span_pending[k] is not None:
                self._wait(self.span_pending[k])
                self.span_pending[k] = None
