"""PipelinedRollout: one env batch as k sub-batches on k HIP streams.

A step kernel over 65 536 envs has a head (every workgroup loads, then computes, then stores at the same time) and a
tail (the last workgroups of the grid run on a nearly empty chip); at BASELINE config 4 the two are about a sixth of the
launch (profiles/archive/NOTES.md, the two-stream probe of round 2: the launch time is 0.05 ms + 0.255 ms per 65 536 envs).  Envs are independent
(the reference steps them in separate OS processes, onpolicy/envs/env_wrappers.py:951-1026), so consecutive steps of
DIFFERENT envs need no ordering: the batch is split into k contiguous sub-batches, each a ``RolloutEngine`` on its own
stream with ``env_offset`` so that the union is the same set of envs with the same random streams, and the tail of one
sub-batch's kernel overlaps the head of another's next step.  What a sub-batch computes is bit-identical to the same
envs inside one big engine.

This is the device-side form of the alternating / double-buffered sampler: the learner consumes sub-batch j while
sub-batch j + 1 steps.  ``step`` only orders a sub-batch behind the caller's stream (its actions) and behind its own
previous step; ``join`` orders the caller's stream behind all sub-batches.
"""
import torch

from .config import EnvConfig
from .engine import RolloutEngine


class PipelinedRollout:
    def __init__(self, cfg, n_envs, k=2, device='cuda:0', seed=0, env_offset=0, **engine_kwargs):
        if not isinstance(cfg, EnvConfig):
            cfg = EnvConfig.from_args(cfg)
        if k < 1 or n_envs % k:
            raise ValueError('n_envs (%d) must be a multiple of the number of sub-batches (%d)' % (n_envs, k))
        self.cfg, self.n_envs, self.k, self.n_sub = cfg, int(n_envs), int(k), int(n_envs) // int(k)
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.streams = [torch.cuda.Stream(self.device) for _ in range(self.k)]
        self.engines = []
        for j, s in enumerate(self.streams):
            with torch.cuda.stream(s):   # the engine initialises its state on the stream that will step it
                self.engines.append(RolloutEngine(cfg, self.n_sub, device=self.device, seed=seed,
                                                  env_offset=env_offset + j * self.n_sub, **engine_kwargs))
        self._ready = [torch.cuda.Event() for _ in range(self.k)]
        self._fed = torch.cuda.Event()

    def _split(self, x):
        if isinstance(x, (list, tuple)):
            if len(x) != self.k:
                raise ValueError('expected %d per-sub-batch tensors, got %d' % (self.k, len(x)))
            return list(x)
        if x.shape[0] != self.n_envs:
            raise ValueError('leading dimension %d != n_envs %d' % (x.shape[0], self.n_envs))
        return [x[j * self.n_sub:(j + 1) * self.n_sub] for j in range(self.k)]   # contiguous leading-dimension slices

    def _each(self, fn):
        cur = torch.cuda.current_stream(self.device)
        self._fed.record(cur)
        for j, (e, s) in enumerate(zip(self.engines, self.streams)):
            s.wait_event(self._fed)   # whatever the caller's stream produced for this call (actions) is ready
            with torch.cuda.stream(s):
                fn(j, e)
                self._ready[j].record(s)

    def reset(self):
        self._each(lambda j, e: e.reset())

    def step(self, actions, auto_reset=True):
        """One step of every sub-batch, each on its own stream (asynchronous; see ``join``).  ``actions``: one
        (n_envs, N) int32 / (n_envs, N, 5) float tensor, or a list of k per-sub-batch tensors."""
        parts = self._split(actions)

        def one(j, e):
            if isinstance(parts[j], torch.Tensor) and parts[j].is_cuda:
                parts[j].record_stream(self.streams[j])   # allocated on the caller's stream, read on this one
            e.step(parts[j], auto_reset=auto_reset)
        self._each(one)

    def rollout(self, action_tapes, mode=None, rings=None):
        """``RolloutEngine.rollout`` of every sub-batch on its own stream: ``action_tapes`` = one contiguous (T, n_envs / k, N)
        int32 device tape per sub-batch (a slice of a (T, n_envs, N) tape along the env axis is not contiguous: split the
        tape once, ``split_tape``).  With spans (the engines' default mode) a sub-batch's run of steps is one launch, and the
        launch boundaries of one sub-batch -- the episode-ending step between two runs, where the chip drains and refills --
        fall into the other's steady state: 10 agents x 65 536 envs 0.181 -> 0.163 ms per step (profiles/archive/r3_notes.md).
        ``rings``: one ``OutputRing`` per sub-batch (``new_rings``) -- step t of every sub-batch goes to its time slot t."""
        if len(action_tapes) != self.k:
            raise ValueError('expected %d per-sub-batch tapes, got %d' % (self.k, len(action_tapes)))

        def one(j, e):
            action_tapes[j].record_stream(self.streams[j])
            e.rollout(action_tapes[j], mode=mode, ring=rings[j] if rings is not None else None)
        self._each(one)

    def new_rings(self, slots, like=None):
        """One ``OutputRing`` of ``slots`` time slots per sub-batch.  ``like``: a ring of the WHOLE batch (an engine of n_envs envs
        made it) -- sub-batch j's slots are then its envs' part of that ring's slots, so the trajectory of the whole batch ends up
        in one set of (T, n_envs, ...) arrays whichever way it was stepped."""
        from .engine import OutputRing
        if like is None:
            return [OutputRing(e, slots) for e in self.engines]
        if like.slots != slots or like.engine.n_envs != self.n_envs:
            raise ValueError('new_rings(like=...): the other ring has %d slots of %d envs, wanted %d of %d'
                             % (like.slots, like.engine.n_envs, slots, self.n_envs))
        return [OutputRing(e, slots, like=like, env_range=(j * self.n_sub, self.n_sub)) for j, e in enumerate(self.engines)]

    def split_tape(self, action_tape):
        """(T, n_envs, N) -> k contiguous (T, n_envs / k, N) tapes, sub-batch j holding the envs [j n_envs / k, (j + 1) n_envs / k)."""
        return [action_tape[:, j * self.n_sub:(j + 1) * self.n_sub].contiguous() for j in range(self.k)]

    def join(self, j=None):
        """Order the caller's stream behind sub-batch j's last step (all sub-batches if None)."""
        cur = torch.cuda.current_stream(self.device)
        for i in (range(self.k) if j is None else (j,)):
            cur.wait_event(self._ready[i])

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def outputs(self, name):
        """The k per-sub-batch tensors of output ``name`` (obs, reward, done, info, node_obs, adj_env); sub-batch j holds
        the envs [j n_envs / k, (j + 1) n_envs / k)."""
        return [getattr(e.outs, name) for e in self.engines]

    def gather(self, name):
        """Output ``name`` of the whole batch as one tensor (a copy: ``join`` first orders it behind the steps)."""
        self.join()
        return torch.cat(self.outputs(name), dim=0)

    def close(self):
        for e in self.engines:
            e.close()
