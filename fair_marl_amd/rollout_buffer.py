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
"""
import torch


class DeviceRolloutBuffer(object):
    def __init__(self, engine, episode_length=None):
        self.engine = eng = engine
        cfg = eng.cfg
        self.T = T = int(episode_length or cfg.episode_length)
        n, N, E, D, F = eng.n_envs, cfg.N, cfg.E, cfg.obs_dim, cfg.node_feat
        dev = eng.device
        z = lambda *shape, dtype=torch.float32: torch.zeros(*shape, dtype=dtype, device=dev)  # noqa: E731
        self.obs = z(T + 1, n, N, D)
        self.node_obs = z(T + 1, n, N, E, F)
        self.adj_env = z(T + 1, n, E, E)
        self.rewards = z(T, n, N, 1)
        self.dones = z(T, n, N, dtype=torch.uint8)
        self.masks = torch.ones(T + 1, n, N, 1, dtype=torch.float32, device=dev)
        self.active_masks = torch.ones(T + 1, n, N, 1, dtype=torch.float32, device=dev)
        self.agent_id = torch.arange(N, dtype=torch.int32, device=dev).view(1, 1, N, 1).expand(T + 1, n, N, 1)
        self.share_agent_id = torch.arange(N, dtype=torch.int32, device=dev).view(1, 1, 1, N).expand(T + 1, n, N, N)
        self.info_planes = z(T + 1, 14, n, N) if eng.emit_info else None   # one set of info planes per slot (the last step's: process_infos)
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
        done = self.dones[t].to(torch.bool)                          # (n, N)
        done_env = done.all(dim=1, keepdim=True)                     # graph_mpe_runner.py:444
        self.masks[t + 1] = (~done).to(torch.float32).unsqueeze(-1)  # :452-458
        self.active_masks[t + 1] = (~(done & ~done_env)).to(torch.float32).unsqueeze(-1)  # :459-465
        self.step = t + 1
        return self._sets[t + 1]

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
        done = self.dones[t0:t0 + T].to(torch.bool)                         # (T, n, N)
        done_env = done.all(dim=2, keepdim=True)
        self.masks[t0 + 1:t0 + T + 1] = (~done).to(torch.float32).unsqueeze(-1)
        self.active_masks[t0 + 1:t0 + T + 1] = (~(done & ~done_env)).to(torch.float32).unsqueeze(-1)
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
        for name in ('obs', 'node_obs', 'adj_env', 'masks', 'active_masks'):
            buf = getattr(self, name)
            buf[0].copy_(buf[-1])
        self.step = 0


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
