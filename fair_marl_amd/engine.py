"""RolloutEngine: device-resident batch of environments driven through the C-ABI of libfmarl.

One engine = one GPU = ``n_envs`` independent worlds (the reference runs one Python ``World``
per OS process: onpolicy/envs/env_wrappers.py:951-1026).  All buffers are PyTorch-ROCm tensors
owned here and handed to the library as raw pointers; PyTorch is only the allocator / stream
provider.  Outputs stay on the device (float32); ``adj`` is stored once per env and exposed per
agent as a stride-0 ``expand`` (the reference returns N identical copies,
multiagent/custom_scenarios/navigation_graph.py:1033).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .config import EnvConfig

_TORCH_DT = {_lib.DTYPE_F64: torch.float64, _lib.DTYPE_I32: torch.int32, _lib.DTYPE_I8: torch.int8}


class OutputSet(object):
    """obs / reward / done / info tensors of one step + the FmarlOutputs struct pointing at them."""
    __slots__ = ('obs', 'reward', 'done', 'info', 'info_planes', 'node_obs', 'adj_env', 'edge_nnz', 'graph_record', 'c', 'c_noinfo')


class RolloutEngine:
    def __init__(self, cfg, n_envs, device='cuda:0', seed=0, env_offset=0, emit_info=True, async_reset=True,
                 emit_graph=True, tune_placement=None, count_edges=False, emit_graph_record=False, envs_per_workgroup=0):
        if not isinstance(cfg, EnvConfig):
            cfg = EnvConfig.from_args(cfg)
        cfg.validate()
        self.cfg, self.n_envs = cfg, int(n_envs)
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError('fair_marl_amd needs an AMD GPU (torch.cuda.is_available() is False); '
                               'there is no CPU fallback for the rollout path')
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        # async_reset: stage the next episode (placement + fair assignment) on a side stream while the
        # current one runs; same results, the reset leaves the critical path (see include/fmarl.h)
        # envs_per_workgroup: launch geometry override (0 = the library's choice); results never depend on it
        self.c = cfg.to_c(n_envs, seed=seed, env_offset=env_offset, async_reset=async_reset, envs_per_workgroup=envs_per_workgroup)
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):   # the handle's side stream, events and kernel attributes belong to THIS device
            _lib.check(self.lib.fmarl_create(C.byref(self.c), C.byref(self.handle)), 'fmarl_create')
        nbytes = self.lib.fmarl_state_bytes(C.byref(self.c))
        n, N, E = self.n_envs, cfg.N, cfg.E
        D, F = cfg.obs_dim, cfg.node_feat
        with torch.cuda.device(self.device):
            self.state = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            self._state_ptr = self.state.data_ptr()
            # emit_graph=False (the reference's non-graph MPEEnv, MPE_env.py:21-53): node_obs / adj are never
            # computed -- the kernels skip the whole emission when handed NULL pointers
            # (arrays of 64 MiB and more are made of hipMemCreate pieces -- alloc_time_slots: at 10 agents x 65 536 envs a launch streams
            # into those 13 % faster than into one plain allocation, profiles/r4_notes.md)
            self.node_obs = alloc_time_slots(self.lib, self.device, (1, n, N, E, F), zero=True, single=True)[0][0] if emit_graph else None
            self.adj_env = alloc_time_slots(self.lib, self.device, (1, n, E, E), zero=True, single=True)[0][0] if emit_graph else None
            self.agent_id = torch.arange(N, device=self.device).view(1, N, 1).expand(n, N, 1)
        self.emit_info, self.emit_graph = emit_info, emit_graph
        # count_edges: the adj emission also counts every env's policy edges (process_adj then never reads adj back)
        self.count_edges = bool(count_edges and emit_graph)
        # emit_graph_record: scenarios whose node features depend on per-step scenario state (fair_graph_formation) also
        # write the compact per-step record a learner on another GPU rebuilds node_obs from (step_record_words per agent)
        self.step_record_words = int(self.lib.fmarl_step_record_words(C.byref(self.c)))
        self.emit_graph_record = bool(emit_graph_record and self.step_record_words)
        self.placement_ms = None
        # tune_placement = k > 1: time k candidate allocations of node_obs / adj and keep the fastest pair (rounds 2-3: for ONE output
        # set that every step rewrites, some boxes run 10 % apart between allocation pairs).  Off by default since round 4: a
        # rollout that keeps its steps writes time slots (OutputRing / DeviceRolloutBuffer), whose interleaved memory does not depend
        # on the luck of one allocation
        if tune_placement is None:
            tune_placement = 0
        if tune_placement and tune_placement > 1:
            self._tune_output_placement(int(tune_placement))
        self._default_graph = (self.node_obs, self.adj_env)
        self.outs = self.new_output_set()
        self.use_outputs(self.outs)
        self._fields = {}
        shapes = self._field_shapes()
        for fid, name in enumerate(_lib.FIELD_NAMES):
            off, cnt, dt = C.c_size_t(), C.c_size_t(), C.c_int()
            _lib.check(self.lib.fmarl_state_field(C.byref(self.c), fid, C.byref(off), C.byref(cnt), C.byref(dt)),
                       'fmarl_state_field')
            if cnt.value == 0:
                if 0 in shapes[name]:   # empty field of this config (e.g. no walls), not a foreign scenario's field
                    self._fields[name] = torch.zeros(shapes[name], dtype=_TORCH_DT[dt.value], device=self.device)
                continue
            esz = _lib.DTYPE_BYTES[dt.value]
            view = self.state[off.value: off.value + cnt.value * esz].view(_TORCH_DT[dt.value])
            self._fields[name] = view.view(shapes[name])
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fmarl_init_state(self.handle, self.state.data_ptr(), self._stream()), 'fmarl_init_state')

    def _tune_output_placement(self, candidates, launches=5, min_spread=0.03):
        """Pick the fastest (node_obs allocation, adj allocation) pair out of a few -- bounded, and only where it pays.

        How fast the store stream of the emission runs depends, on SOME boxes, on which physical pages the two allocations
        happened to get, and on the pair rather than on either buffer: 1.36 to 1.54 ms per emission launch over the pairs
        of one process at BASELINE config 3, repeatable to 0.2 % for a given pair; on most boxes all pairs agree within 2 %.
        ``profiles/archive/r2_placement_tcc.md`` has the counters: the L2 -> fabric write requests are spread evenly over the 128
        TCC channels for fast and slow pairs alike (no aliasing the kernel could undo); the slow pair back-pressures a few
        channels harder (DRAM credit stalls, max over channels +12 %), a property of the physical page placement that
        neither the kernel nor the caller controls.  So the remedy is to measure: every pair of 3 node_obs x ``candidates``
        adj allocations is timed with the pure emission kernel (fmarl_rebuild_graph: writes node_obs / adj only, touches no
        env state) and the losers are freed.  Bounds: the probe is skipped, with a log line, when its transient allocations
        would not fit beside what already owns the HBM (e.g. a DeviceRolloutBuffer), and it stops after the first row of
        pairs when those differ by less than ``min_spread`` (the common box).  ``placement_ms`` keeps the timing matrix (row 0
        / column 0 = the allocations the engine started with)."""
        import logging
        log = logging.getLogger('fair_marl_amd')
        cfg, n, dev = self.cfg, self.n_envs, self.device
        if cfg.scenario_name != 'navigation_graph' or self.node_obs is None:
            log.info('output placement probe skipped: it times the navigation_graph emission kernel (scenario %s)', cfg.scenario_name)
            return
        node_bytes = self.node_obs.numel() * 4
        adj_bytes = self.adj_env.numel() * 4
        with torch.cuda.device(dev):
            free, _ = torch.cuda.mem_get_info(dev)
            first_row = (candidates - 1) * adj_bytes + (64 << 20)
            if first_row > free // 2:   # never take more than half of what is left
                log.info('output placement probe skipped: %.1f GB of candidates would not fit beside the %.1f GB in use',
                         first_row / 1e9, (torch.cuda.mem_get_info(dev)[1] - free) / 1e9)
                return
            obs = torch.zeros(n, cfg.N, cfg.obs_dim, dtype=torch.float32, device=dev)
            rec = torch.zeros(n, self.episode_record_words, dtype=torch.int32, device=dev)
            nodes, adjs = [self.node_obs], [self.adj_env] + [torch.empty_like(self.adj_env) for _ in range(candidates - 1)]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def time_row(node):
                row = []
                for adj in adjs:
                    for _ in range(2):
                        self.rebuild_graph(obs, rec, node_obs=node, adj_env=adj)
                    e0.record()
                    for _ in range(launches):
                        self.rebuild_graph(obs, rec, node_obs=node, adj_env=adj)
                    e1.record()
                    e1.synchronize()
                    row.append(e0.elapsed_time(e1) / launches)
                return row
            times = [time_row(nodes[0])]
            spread = max(times[0]) / min(times[0]) - 1.0
            free, _ = torch.cuda.mem_get_info(dev)
            if spread < min_spread:
                log.info('output placement: the %d adj candidates differ by %.1f %% on this box, probe stopped', candidates, 100 * spread)
            elif 2 * node_bytes > free // 2:
                log.info('output placement: node_obs candidates (%.1f GB) skipped, %.1f GB free', 2 * node_bytes / 1e9, free / 1e9)
            else:
                for _ in range(2):
                    nodes.append(torch.empty_like(self.node_obs))
                    times.append(time_row(nodes[-1]))
            bi, bj = min(((i, j) for i in range(len(nodes)) for j in range(len(adjs))), key=lambda ij: times[ij[0]][ij[1]])
            self.node_obs, self.adj_env = nodes[bi], adjs[bj]
            self.node_obs.zero_()
            self.adj_env.zero_()
            del nodes, adjs, obs, rec
            torch.cuda.empty_cache()   # hand the losing allocations back to the driver
        self.placement_ms = times

    def new_output_set(self, obs=None, reward=None, done=None, node_obs=None, adj_env=None, graph_record=None, info_planes=None,
                       edge_nnz=None):
        """A set of per-step output buffers.  By default node_obs / adj are shared by all sets (large,
        consumed before the next step) while obs / reward / done / info are per set, so a set can still
        be read (e.g. by an in-flight RCCL gather, see sharding.py) while the next step writes another.
        Any buffer can be supplied by the caller (e.g. a slot of a DeviceRolloutBuffer: zero-copy insert)."""
        n, N, D = self.n_envs, self.cfg.N, self.cfg.obs_dim
        with torch.cuda.device(self.device):
            mk = lambda t, shape, dt: t if t is not None else torch.zeros(shape, dtype=dt, device=self.device)
            o = OutputSet()
            o.obs = mk(obs, (n, N, D), torch.float32)
            o.reward = mk(reward, (n, N), torch.float32)
            o.done = mk(done, (n, N), torch.uint8)
            # field-major planes on the device (coalesced stores); exposed as an (n, N, K) view
            o.info_planes = info_planes if info_planes is not None else (
                torch.zeros(_lib.INFO_WIDTH, n, N, dtype=torch.float32, device=self.device) if self.emit_info else None)
            if o.info_planes is not None and (tuple(o.info_planes.shape) != (_lib.INFO_WIDTH, n, N) or o.info_planes.dtype != torch.float32
                                              or not o.info_planes.is_contiguous()):
                raise ValueError('info_planes must be a contiguous float32 tensor of shape %s' % ((_lib.INFO_WIDTH, n, N),))
            o.info = o.info_planes.permute(1, 2, 0) if o.info_planes is not None else None
            o.node_obs = node_obs if node_obs is not None else self._default_graph[0]
            o.adj_env = adj_env if adj_env is not None else self._default_graph[1]
            o.edge_nnz = edge_nnz if edge_nnz is not None else (torch.zeros(n, dtype=torch.int32, device=self.device) if self.count_edges else None)
            o.graph_record = graph_record
            if o.graph_record is None and self.emit_graph_record:
                o.graph_record = torch.zeros(n, N, self.step_record_words, dtype=torch.int32, device=self.device)
        E, F = self.cfg.E, self.cfg.node_feat
        for t, shape, dt in ((o.obs, (n, N, D), torch.float32), (o.reward, (n, N), torch.float32), (o.done, (n, N), torch.uint8),
                             (o.node_obs, (n, N, E, F), torch.float32), (o.adj_env, (n, E, E), torch.float32)):
            if t is None:
                continue
            if tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError('output buffer must be a contiguous %s tensor of shape %s on %s' % (dt, shape, self.device))
        o.c = _lib.FmarlOutputs(o.obs.data_ptr(), o.node_obs.data_ptr() if o.node_obs is not None else None,
                                o.adj_env.data_ptr() if o.adj_env is not None else None,
                                o.reward.data_ptr(), o.done.data_ptr(),
                                o.info_planes.data_ptr() if o.info_planes is not None else None,
                                o.edge_nnz.data_ptr() if o.edge_nnz is not None else None,
                                o.graph_record.data_ptr() if o.graph_record is not None else None)
        o.c_noinfo = _lib.FmarlOutputs(o.c.obs, o.c.node_obs, o.c.adj, o.c.reward, o.c.done, None, o.c.edge_nnz, o.c.graph_record)
        if o.graph_record is not None and (tuple(o.graph_record.shape) != (n, N, self.step_record_words) or o.graph_record.dtype != torch.int32
                                           or not o.graph_record.is_contiguous()):
            raise ValueError('graph_record must be a contiguous int32 tensor of shape %s' % ((n, N, self.step_record_words),))
        return o

    def use_outputs(self, out_set):
        """Select the output set the next reset / step calls write into."""
        self.outs = out_set
        self._outs_ref = C.byref(out_set.c)   # (built once per set: the step path of a launch-bound batch counts microseconds)
        self._outs_noinfo_ref = C.byref(out_set.c_noinfo)
        self.obs, self.reward, self.done, self.info = out_set.obs, out_set.reward, out_set.done, out_set.info
        self.node_obs, self.adj_env, self.graph_record = out_set.node_obs, out_set.adj_env, out_set.graph_record

    def _field_shapes(self):
        n, c = self.n_envs, self.cfg
        N, L, O, W = c.num_agents, c.num_landmarks, c.num_obstacles, c.num_walls
        return dict(agent_pos=(n, N, 2), agent_vel=(n, N, 2), p_dist=(n, N), landmark_pos=(n, L, 2),
                    obstacle_pos=(n, O, 2), wall_axis=(n, W), wall_e0=(n, W), wall_e1=(n, W), wall_orient=(n, W),
                    wall_length=(n,), goal_match=(n, N), dists_to_goal=(n, N), times_required=(n, N),
                    dist_left=(n, N), num_obst_coll=(n, N), num_agent_coll=(n, N), min_time=(n, N),
                    cur_step=(n,), episode=(n,), slot_pos=(n, N, 2), slot_occ=(n, N), slot_delta=(n, N),
                    formation_done=(n, N), goal_occ=(n, N), goal_history=(n, N), goal_reached=(n, N), status=(n, N),
                    reset_flag=(n,), stage_agent_pos=(n, N, 2), stage_landmark_pos=(n, L, 2),
                    stage_obstacle_pos=(n, O, 2), stage_wall_axis=(n, W), stage_wall_orient=(n, W),
                    stage_goal_match=(n, N), stage_valid=(n,), stage_need=(n,), place_fails=(n,), stage_place_fails=(n,),
                    internal_match_dual=(n, N), internal_rot_table=(N, 2))

    @property
    def envs_per_workgroup(self):
        """Envs per workgroup of this engine's step launches (FmarlConfig.envs_per_workgroup or the library's choice)."""
        return int(self.lib.fmarl_envs_per_workgroup(self.handle))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ state access (parity harness)
    def field(self, name):
        """Device tensor view of one state field (see include/fmarl.h FMARL_F_*)."""
        return self._fields[name]

    _NP_DT = {torch.float64: np.float64, torch.int32: np.int32, torch.int8: np.int8}

    def get_state(self):
        """All scenario state fields as NumPy arrays in the reference's shapes (fmarl_get_state)."""
        out = {}
        with torch.cuda.device(self.device):
            torch.cuda.current_stream(self.device).synchronize()
            for k, t in self._fields.items():
                if k == 'reset_flag' or k.startswith(('stage_', 'internal_')):
                    continue
                a = np.empty(tuple(t.shape), dtype=self._NP_DT[t.dtype])
                if a.size:
                    _lib.check(self.lib.fmarl_get_state(self.handle, self.state.data_ptr(), _lib.FIELD_NAMES.index(k),
                                                        a.ctypes.data, self._stream()), 'fmarl_get_state')
                out[k] = a
            torch.cuda.current_stream(self.device).synchronize()
        return out

    def set_state(self, state):
        """Inject state fields (parity harness; fmarl_set_state, which also tells the handle the state changed)."""
        with torch.cuda.device(self.device):
            keep = []
            for k, v in state.items():
                if k not in self._fields:
                    continue
                t = self._fields[k]
                a = np.ascontiguousarray(np.asarray(v).astype(self._NP_DT[t.dtype]).reshape(tuple(t.shape)))
                keep.append(a)
                if a.size:
                    _lib.check(self.lib.fmarl_set_state(self.handle, self.state.data_ptr(), _lib.FIELD_NAMES.index(k),
                                                        a.ctypes.data, self._stream()), 'fmarl_set_state')
            torch.cuda.current_stream(self.device).synchronize()   # host sources must outlive the copies
        _lib.check(self.lib.fmarl_state_changed(self.handle), 'fmarl_state_changed')

    def placement_exhausted(self):
        """int32 device tensor (n,): entities of each env's current episode that the reset placed although all 10 000
        rejection-sampling draws collided with something (include/fmarl.h FMARL_F_PLACE_FAILS).  The reference loops until
        a free spot turns up (navigation_graph.py:389-457, :472-535) -- forever in a world too crowded to have one; the
        bounded loop accepts the last draw instead and says so here.  All zeros for any sane configuration."""
        return self._fields['place_fails']

    # ------------------------------------------------------------------ hot path
    @property
    def adj(self):
        """(n, N, E, E) stride-0 view of the per-env distance matrix."""
        if self.adj_env is None:
            return None
        n, E = self.n_envs, self.cfg.E
        return self.adj_env.view(n, 1, E, E).expand(n, self.cfg.N, E, E)

    def reset(self, env_mask=None):
        """MultiAgentGraphEnv.reset for all (or the masked) envs -> (obs, agent_id, node_obs, adj)."""
        mask_ptr = None
        if env_mask is not None:
            env_mask = torch.as_tensor(env_mask).to(device=self.device, dtype=torch.uint8).contiguous()
            mask_ptr = env_mask.data_ptr()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fmarl_reset(self.handle, self.state.data_ptr(), mask_ptr, C.byref(self.outs.c),
                                            self._stream()), 'fmarl_reset')
        return self.obs, self.agent_id, self.node_obs, self.adj

    def step(self, actions, auto_reset=True, emit_info=True):
        """One env step for every env.  ``actions``: int tensor (n, N) of indices 0..4 or float tensor
        (n, N, 5) in the reference's one-hot / continuous form.  Returns device tensors
        (obs, agent_id, node_obs, adj, reward, done, info).

        ``emit_info=False`` skips the 14 info planes of THIS step (56 bytes per agent-step of pure output; the world state
        the reference's info_callback updates -- arrival times, collision counters, ... -- is kept either way).  The
        reference's runner only reads the infos of an episode's last step (onpolicy/runner/shared/base_runner.py:197-276,
        graph_mpe_runner.py:146); the returned ``info`` tensor then still holds the planes of the last step that wrote them."""
        a = actions if isinstance(actions, torch.Tensor) else torch.as_tensor(np.asarray(actions))
        n, N = self.n_envs, self.cfg.N
        idx_ptr = vec_ptr = None
        if a.dim() == 2:
            if a.shape[0] != n or a.shape[1] != N:
                raise ValueError('action index tensor must have shape (%d, %d), got %s' % (n, N, tuple(a.shape)))
            if a.dtype != torch.int32 or a.device != self.device or not a.is_contiguous():
                a = a.to(device=self.device, dtype=torch.int32).contiguous()
            idx_ptr = a.data_ptr()
        else:
            if tuple(a.shape) != (n, N, 5):
                raise ValueError('action tensor must have shape (%d, %d, 5), got %s' % (n, N, tuple(a.shape)))
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
            vec_ptr = a.data_ptr()
        # (the library switches to the handle's device itself; torch only supplies the stream of THAT device)
        rc = self.lib.fmarl_step(self.handle, self._state_ptr, idx_ptr, vec_ptr, self._outs_ref if emit_info else self._outs_noinfo_ref, int(auto_reset),
                                 torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            _lib.check(rc, 'fmarl_step')
        self._last_actions = a  # keep alive until the stream has consumed it
        return self.obs, self.agent_id, self.node_obs, self.adj, self.reward, self.done, self.info

    # ------------------------------------------------------------------ exported pieces
    def cost_matrix(self, agent_pos, goal_pos):
        """cdist(agent_pos, goal_pos): f64 (n, N, 2), (n, L, 2) -> (n, N, L) on the device."""
        ap = torch.as_tensor(agent_pos).to(device=self.device, dtype=torch.float64).contiguous()
        gp = torch.as_tensor(goal_pos).to(device=self.device, dtype=torch.float64).contiguous()
        n, N, L = ap.shape[0], ap.shape[1], gp.shape[1]
        out = torch.empty(n, N, L, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fmarl_cost_matrix(ap.data_ptr(), gp.data_ptr(), out.data_ptr(), n, N, L,
                                                  self._stream()), 'fmarl_cost_matrix')
        return out

    def lexifair(self, costs):
        """solve_fair_assignment for a batch of cost matrices f64 (n, N, N) -> perm int32 (n, N)."""
        c = torch.as_tensor(costs).to(device=self.device, dtype=torch.float64).contiguous()
        n, N = c.shape[0], c.shape[1]
        perm = torch.empty(n, N, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fmarl_lexifair(c.data_ptr(), perm.data_ptr(), n, N, self._stream()), 'fmarl_lexifair')
        return perm

    def update_graph(self, adj_env=None):
        """Scenario.update_graph (navigation_graph.py:1037-1056): (edge_index int32 (n, 2, E*E) padded with -1,
        edge_weight (n, E*E), nnz (n)).  Without an argument the edges come from the CURRENT world state exactly as the
        reference computes them -- float64 distances, ``<=`` in float64, float64 weights (call it where
        MultiAgentGraphEnv.step does: before the step).  With a float32 ``adj_env`` (n, E, E) (e.g. a stored rollout
        slot) the same rule is applied to that matrix in float32."""
        if adj_env is None:
            n, E = self.n_envs, self.cfg.E
            ei = torch.empty(n, 2, E * E, dtype=torch.int32, device=self.device)
            ew = torch.empty(n, E * E, dtype=torch.float64, device=self.device)
            nnz = torch.empty(n, dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.fmarl_update_graph_state(self.handle, self.state.data_ptr(), ei.data_ptr(), ew.data_ptr(),
                                                             nnz.data_ptr(), self._stream()), 'fmarl_update_graph_state')
            return ei, ew, nnz
        adj = torch.as_tensor(adj_env).to(self.device, torch.float32).contiguous()
        n, E = adj.shape[0], adj.shape[1]
        ei = torch.empty(n, 2, E * E, dtype=torch.int32, device=self.device)
        ew = torch.empty(n, E * E, dtype=torch.float32, device=self.device)
        nnz = torch.empty(n, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fmarl_update_graph(adj.data_ptr(), ei.data_ptr(), ew.data_ptr(), nnz.data_ptr(), n, E,
                                                   float(self.cfg.max_edge_dist), self._stream()), 'fmarl_update_graph')
        return ei, ew, nnz

    def process_adj(self, adj_env=None, per_agent=False, strict=True, max_edges=None):
        """Policy-side edge list (reference onpolicy/algorithms/utils/gnn.py:307-326 processAdj + the PyG batching of
        :243-253): edges with 0 < adj < max_edge_dist in row-major order, node ids offset by graph index * E.
        ``per_agent=True`` replicates every env's graph N times like the reference batch of (env, agent) graphs; the
        default emits each env's graph once.  Returns (edge_index int64 (2, cap), edge_attr f32 (cap,), offsets int64
        (n_graphs + 1)); ``offsets[-1]`` is the number of edges.

        Without ``adj_env``, on an engine created with ``count_edges=True``, this is the fused path of SURVEY section 8 f-3:
        the counts come from the adj emission of the last step / reset, the prefix sum runs on the device and the edges
        are rebuilt from the world state -- adj is never read back.  ``max_edges=None`` sizes the result exactly, which
        takes the one host read of ``offsets[-1]`` any exact-size tensor needs; with ``max_edges=K`` the buffers hold K
        edges (extra ones are dropped, ``offsets`` still counts them) and nothing synchronises with the host.
        With ``adj_env`` (e.g. a stored rollout slot), ``strict=False`` (update_graph's ``<=``) or without
        ``count_edges`` the counts come from one pass over the float32 matrix and the edges from a second one."""
        n = self.n_envs if adj_env is None else int(adj_env.shape[0])
        E, reps = self.cfg.E, (self.cfg.N if per_agent else 1)
        thr = float(self.cfg.max_edge_dist)
        fused = adj_env is None and strict and self.outs.edge_nnz is not None
        with torch.cuda.device(self.device):
            if fused:
                nnz = self.outs.edge_nnz
            else:
                adj = self.adj_env if adj_env is None else torch.as_tensor(adj_env).to(self.device, torch.float32).contiguous()
                E = adj.shape[1]
                nnz = torch.empty(n, dtype=torch.int32, device=self.device)
                _lib.check(self.lib.fmarl_edge_count(adj.data_ptr(), nnz.data_ptr(), n, E, thr, 1 if strict else 0, self._stream()),
                           'fmarl_edge_count')
            offsets = torch.empty(n * reps + 1, dtype=torch.int64, device=self.device)
            _lib.check(self.lib.fmarl_edge_offsets(nnz.data_ptr(), n, reps, offsets.data_ptr(), self._stream()), 'fmarl_edge_offsets')
            cap = int(offsets[-1].item()) if max_edges is None else int(max_edges)
            ei = torch.empty(2, cap, dtype=torch.int64, device=self.device)
            ea = torch.empty(cap, dtype=torch.float32, device=self.device)
            if cap and fused:
                if getattr(self, 'edge_mismatch', None) is None:
                    # device counter of graphs whose state-rebuilt edges did not fill their offsets range exactly (the counts
                    # must be those of the step the state is in); stays 0 in correct use, never read by this call
                    self.edge_mismatch = torch.zeros(1, dtype=torch.int32, device=self.device)
                _lib.check(self.lib.fmarl_edge_fill_state(self.handle, self.state.data_ptr(), offsets.data_ptr(), ei.data_ptr(),
                                                          ea.data_ptr(), cap, reps, self.edge_mismatch.data_ptr(), self._stream()),
                           'fmarl_edge_fill_state')
            elif cap:
                _lib.check(self.lib.fmarl_edge_fill(adj.data_ptr(), offsets.data_ptr(), ei.data_ptr(), ea.data_ptr(), cap, n * reps, reps, E,
                                                    thr, 1 if strict else 0, self._stream()), 'fmarl_edge_fill')
        return ei, ea, offsets

    # name of every info key in env_infos, in the order process_infos fills the dict (base_runner.py:245-275)
    _ENV_INFO_NAMES = (('individual_reward', 'individual_rewards'), ('Time_req_to_goal', 'time_to_goal'),
                       ('Min_time_to_goal', 'min_time_to_goal'), ('Dist_to_goal', 'dist_to_goal'),
                       ('Num_agent_collisions', 'num_agent_collisions'), ('Num_obst_collisions', 'num_obstacle_collisions'),
                       ('Distance_mean', 'distance_mean'), ('Distance_variance', 'distance_variance'),
                       ('Mean_by_variance', 'mean_variance'), ('Dists_traveled', 'dists_traveled'), ('Time_taken', 'time_taken'),
                       ('Formation_dist', 'formation_dist'), ('Time_mean', 'time_mean'), ('Time_stddev', 'time_variance'),
                       ('Time_mean_by_stddev', 'time_mn_by_stddev'))

    def process_infos(self, dt=0.1, reduce='mean'):
        """Episode metrics of the last step's infos, reference onpolicy/runner/shared/base_runner.py:197-276
        ``process_infos``: dict 'agent<i>/<name>' for the 15 names of :245-258, Time_req_to_goal == -1 counted as
        episode_length * dt (:212-215).  ``reduce='mean'`` gives what ``log_env`` (:291-306) logs, the mean over the
        envs, reduced on the device (names whose key the scenario does not emit are left out, like the reference's
        ``len(v) > 0`` test); ``reduce=None`` gives the reference's per-env lists as float64 device tensors (n,)
        (empty for names the scenario does not emit)."""
        from .infos import key_map
        if self.outs.info_planes is None:
            raise RuntimeError('engine was created with emit_info=False')
        N = self.cfg.N
        slots = dict(key_map(self.cfg.scenario_name))
        if reduce is None:
            planes = self.outs.info_planes.to(torch.float64)          # (K, n, N)
            out = {}
            for a in range(N):
                for key, name in self._ENV_INFO_NAMES:
                    if key not in slots:
                        out['agent%d/%s' % (a, name)] = planes.new_empty(0)
                        continue
                    v = planes[slots[key], :, a]
                    if key == 'Time_req_to_goal':
                        v = torch.where(v == -1.0, torch.full_like(v, self.cfg.episode_length * dt), v)
                    out['agent%d/%s' % (a, name)] = v
            return out
        means = torch.empty(_lib.INFO_WIDTH, N, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fmarl_info_means(self.outs.info_planes.data_ptr(), means.data_ptr(), self.n_envs, N,
                                                 float(self.cfg.episode_length * dt), self._stream()), 'fmarl_info_means')
        m = means.cpu().numpy()
        return {'agent%d/%s' % (a, name): float(m[slots[key], a]) for a in range(N) for key, name in self._ENV_INFO_NAMES
                if key in slots}

    _METRIC_KEYS = {'get_fairness_metric': 'Mean_by_variance', 'get_dist_mean': 'Distance_mean', 'get_dist_std': 'Distance_variance',
                    'get_time_fairness': 'Time_mean_by_stddev', 'get_time_mean': 'Time_mean', 'get_time_std': 'Time_stddev'}

    def _metric(self, reader):
        """The ``get_*`` readers of base_runner.py:308-420: per agent the value of the FIRST env (``v[0]``) of one
        env_infos name.  A name the scenario does not emit raises IndexError, as ``v[0]`` on the reference's empty list."""
        from .infos import key_map
        slots = dict(key_map(self.cfg.scenario_name))
        key = self._METRIC_KEYS[reader]
        if key not in slots:
            raise IndexError('%s: scenario %s emits no %r' % (reader, self.cfg.scenario_name, key))
        return [float(x) for x in self.outs.info_planes[slots[key], 0, :].to(torch.float64).cpu()]

    def get_fairness_metric(self): return self._metric('get_fairness_metric')
    def get_dist_mean(self): return self._metric('get_dist_mean')
    def get_dist_std(self): return self._metric('get_dist_std')
    def get_time_fairness(self): return self._metric('get_time_fairness')
    def get_time_mean(self): return self._metric('get_time_mean')
    def get_time_std(self): return self._metric('get_time_std')

    # ------------------------------------------------------------------ cross-GPU hand-off of the graph observation
    @property
    def episode_record_words(self):
        """32-bit words per env of the episode record (goals, landmarks, obstacles, walls)."""
        return int(self.lib.fmarl_episode_record_words(C.byref(self.c)))

    @property
    def episode_started(self):
        """True if the last reset() / step() may have started episodes (host-side, no device access)."""
        return bool(self.lib.fmarl_episode_started(self.handle))

    def pack_episode(self, out=None):
        """Episode record of every env, int32 (n_envs, episode_record_words): what a learner on another GPU needs
        besides the per-step obs to rebuild node_obs / adj (navigation_graph only)."""
        if out is None:
            out = torch.empty(self.n_envs, self.episode_record_words, dtype=torch.int32, device=self.device)
        assert out.is_contiguous() and out.numel() * out.element_size() == self.n_envs * self.episode_record_words * 4
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fmarl_pack_episode(self.handle, self.state.data_ptr(), out.data_ptr(), self._stream()),
                       'fmarl_pack_episode')
        return out

    def rebuild_graph(self, obs, record, node_obs=None, adj_env=None, want_node_obs=True, want_adj=True, step_record=None):
        """graph_observation (reference navigation_graph.py:941-1035, fair_graph_formation.py:810-971) from the gathered
        records; n is the caller's.  navigation_graph: ``obs`` (n, N, D) f32 + the episode ``record`` (n, words).
        fair_graph_formation: the per-step ``step_record`` (n, N, step_record_words) int32 written by the step kernel
        (``emit_graph_record=True``) + the episode record; ``obs`` is not needed (may be None).
        Returns (node_obs (n, N, E, F) | None, adj_env (n, E, E) | None), bit-identical to the sender's."""
        cfg = self.cfg
        if step_record is not None:
            step_record = step_record.to(self.device).contiguous()
            n = step_record.shape[0]
            assert tuple(step_record.shape[1:]) == (cfg.N, self.step_record_words) and step_record.dtype == torch.int32
        if obs is not None:
            obs = torch.as_tensor(obs).to(self.device, torch.float32).contiguous()
            n = obs.shape[0]
            assert tuple(obs.shape[1:]) == (cfg.N, cfg.obs_dim)
        assert record.is_contiguous() and record.device == self.device
        assert record.numel() * record.element_size() == n * self.episode_record_words * 4
        if node_obs is None and want_node_obs:
            node_obs = torch.empty(n, cfg.N, cfg.E, cfg.node_feat, dtype=torch.float32, device=self.device)
        if adj_env is None and want_adj:
            adj_env = torch.empty(n, cfg.E, cfg.E, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fmarl_rebuild_graph_rec(self.handle, obs.data_ptr() if obs is not None else None, record.data_ptr(),
                                                        step_record.data_ptr() if step_record is not None else None, n,
                                                        node_obs.data_ptr() if node_obs is not None else None,
                                                        adj_env.data_ptr() if adj_env is not None else None, self._stream()),
                       'fmarl_rebuild_graph')
        return node_obs, adj_env

    # ------------------------------------------------------------------ launch-bound batches: one graph per episode
    @property
    def phase(self):
        """Steps since the last reset of all envs while they run in lockstep, else -1 (host-side, no device access)."""
        return int(self.lib.fmarl_get_phase(self.handle))

    def capture_steps(self, action_tape, auto_reset=True, lockstep=False):
        """Capture ``len(action_tape)`` consecutive steps into a hipGraph and return it; ``graph.replay()`` then runs them
        with one launch.  For small batches the step kernel takes 10-20 us and the host-side launch path is the larger
        part of a step.  ``action_tape`` (T, n, N) int32 or (T, n, N, 5) float32 is read at replay time: refill it in place
        between replays.  Each step writes the engine's current output set, so a consumer that wants every step passes
        per-step sets of a DeviceRolloutBuffer via ``outputs`` of ``capture_rollout``.

        ``lockstep=True`` bakes the reset decision from the host's mirror of the common step counter (one reset per episode
        in the graph instead of a test per step: BASELINE config 2, 20.5 -> 14.7 us per step); the returned object refuses a
        replay from any other episode phase than the one it was captured at.  With the staged reset (``async_reset=True``,
        the default) the staging of every next episode becomes a forked branch of the graph, beside the episode's step
        kernels, joined by the launch that ends the episode; such a graph holds whole episodes and starts with the first
        step after a reset.  ``lockstep=False`` (default): every captured step carries the device-side auto-reset test, so the
        graph may hold any number of steps and be replayed from any phase of an episode -- these are the synchronous reset's
        launches, and an engine with the staged reset becomes a synchronous one for good.  ``lockstep=None`` picks the lean
        form whenever it is valid (envs in lockstep; with the staged reset also phase 0 and whole episodes)."""
        return self.capture_rollout(action_tape, None, auto_reset, lockstep)

    def _lean_capture_ok(self, steps):
        if self.phase < 0:
            return False
        staged_nav = bool(self.c.flags & _lib.FLAG_ASYNC_RESET) and self.cfg.scenario_name == 'navigation_graph'
        return not staged_nav or (self.phase == 0 and steps % self.cfg.episode_length == 0)

    def capture_rollout(self, action_tape, outputs=None, auto_reset=True, lockstep=False):
        """As ``capture_steps``; step t writes ``outputs[t]`` (an OutputSet, e.g. a time slot of a rollout buffer)."""
        tape = action_tape.to(self.device)
        phase0 = self.phase
        if lockstep is None:
            lockstep = bool(auto_reset) and self._lean_capture_ok(int(tape.shape[0]))
        if lockstep and phase0 < 0:
            raise RuntimeError('lockstep capture needs all envs in lockstep: reset() them first (phase is -1)')
        if lockstep and not self._lean_capture_ok(int(tape.shape[0])):
            raise RuntimeError('lockstep capture with the staged reset holds whole episodes from the first step after a reset '
                               '(phase %d, %d steps, episode_length %d)' % (phase0, tape.shape[0], self.cfg.episode_length))
        import gc
        gc.collect()   # nothing may be destroyed (streams, events, other graphs) while this stream captures
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        mode = (2 if lockstep else 1) if auto_reset else 0
        with torch.cuda.device(self.device), torch.cuda.graph(graph):
            for t in range(tape.shape[0]):
                if outputs is not None:
                    self.use_outputs(outputs[t])
                self.step(tape[t], auto_reset=mode)
        graph._fmarl_keep = (tape, outputs)   # the captured launches point into these
        if not lockstep:
            return graph
        _lib.check(self.lib.fmarl_set_phase(self.handle, phase0), 'fmarl_set_phase')   # nothing ran during the capture
        return _LockstepGraph(self, graph, phase0, int(tape.shape[0]))

    # a step kernel over fewer agents than this is shorter than the host-side launch path of one fmarl_step call
    GRAPH_BELOW_AGENTS = 1 << 16

    def step_span(self, action_tape, strides=None):
        """``len(action_tape)`` auto-resetting steps from a device tape -- (T, n, N) int32 action indices or (T, n, N, 5) float32
        action vectors (the reference's one-hot / continuous form, as for ``step``) -- through ``fmarl_step_span``: the steps
        between episode ends go out as ONE launch in which every workgroup walks its own envs through time (envs never
        interact; no per-step launch, no per-step head and tail of the grid), the step that ends an episode as a launch of
        its own (nav_fairassign_fairrew_formation_graph resets its ended envs inside the step: the whole tape is one
        launch).  Same results as T ``step`` calls, bit for bit.  By default every step writes the engine's current output
        set (what is left in it are the last step's outputs); ``strides`` = dict of per-step element strides for 'obs',
        'node_obs', 'adj', 'reward', 'done', 'info', 'edge_nnz', 'graph_record' makes step t write the set's buffers shifted
        by t strides -- the time slots of a rollout buffer laid out (T, n, ...) (``DeviceRolloutBuffer.insert_span``)."""
        tape = action_tape
        shape = (self.n_envs, self.cfg.N)
        as_idx = tape.dtype == torch.int32 and tape.dim() == 3 and tuple(tape.shape[1:]) == shape
        as_vec = tape.dtype == torch.float32 and tape.dim() == 4 and tuple(tape.shape[1:]) == shape + (5,)
        if not (as_idx or as_vec) or tape.device != self.device or not tape.is_contiguous():
            raise ValueError('action tape must be a contiguous device tensor, int32 of shape (T, %d, %d) or float32 of shape (T, %d, %d, 5)'
                             % (shape + shape))
        st = strides or {}
        span = _lib.FmarlSpan(*[int(st.get(k, 0)) for k in ('obs', 'node_obs', 'adj', 'reward', 'done', 'info', 'edge_nnz', 'graph_record')],
                              self.n_envs * self.cfg.N * (1 if as_idx else 5))
        rc = self.lib.fmarl_step_span(self.handle, self._state_ptr, tape.data_ptr() if as_idx else None, tape.data_ptr() if as_vec else None,
                                      int(tape.shape[0]), self._outs_ref, C.byref(span),
                                      torch.cuda.current_stream(self.device).cuda_stream)
        if rc:
            _lib.check(rc, 'fmarl_step_span')
        self._last_actions = tape

    def rollout(self, action_tape, mode=None, use_graph=None, ring=None):
        """Run ``len(action_tape)`` auto-resetting steps from a persistent device tape (T, n, N) int32: the random-action
        rollout of the reference's throughput runs, or a scripted tape.  Outputs of the last step are in the engine's current
        output set (with ``ring``: the time slot of the last step becomes the current set, in every mode).  ``mode``:

        * ``'span'`` (default): ``step_span`` -- one launch per run of steps between episode ends (10 agents x 65 536 envs:
          0.250 -> 0.199 ms per step; 3 agents x 4 096 envs: 14.5 -> 11.5 us); nav_fairassign_fairrew_formation_graph, whose
          episodes end env by env inside the step: the whole tape as ONE launch (65 536 x 3: 0.059 -> 0.050 ms per step);
        * ``'graph'``: one hipGraph replay per call, captured on first use and cached per (tape tensor, length, output set,
          episode phase) -- valid while the caller refills the same tensor in place; needs envs in lockstep and a phase / length
          the lean capture covers, otherwise the call falls through to ``'eager'`` (always so for
          nav_fairassign_fairrew_formation_graph, whose envs are never in lockstep);
        * ``'eager'``: one ``step`` call per step.

        ``use_graph`` (older spelling): True = 'graph' where valid, False = 'eager'.

        ``ring`` (an ``OutputRing`` of at least ``len(action_tape)`` slots; modes 'span' and 'eager'): step t writes time slot t
        instead of every step overwriting the engine's current output set -- the trajectory exists afterwards, as the
        reference's runner keeps it (onpolicy/envs/env_wrappers.py:988-996 delivers every step's outputs)."""
        T = int(action_tape.shape[0])
        if mode is None and use_graph is not None:
            mode = 'graph' if use_graph else 'eager'
        if mode is None:
            mode = 'span'
        if ring is not None and (T > ring.slots or mode == 'graph'):
            raise ValueError('rollout: the ring holds %d slots, the tape %d steps (and graph replays write one output set)' % (ring.slots, T))
        if T == 0:
            return
        if mode == 'span':
            if ring is not None:
                self.use_outputs(ring.sets[0])
            self.step_span(action_tape, strides=ring.strides if ring is not None else None)
            if ring is not None:   # as the eager mode and DeviceRolloutBuffer.insert_span leave it: the current set is the LAST step's
                self.use_outputs(ring.sets[T - 1])
            return
        if mode == 'graph' and self._lean_capture_ok(T):
            # the cache holds the tape and the output set themselves (not their addresses: an address can be reused by another
            # tensor once the first one is gone, and the old graph would write into freed buffers), at most 8 graphs
            key = (action_tape.data_ptr(), T, self.phase)
            cache = self.__dict__.setdefault('_rollout_graphs', [])
            hit = next((e for e in cache if e[0] == key and e[2] is self.outs and e[1].data_ptr() == action_tape.data_ptr()
                        and e[1].untyped_storage().data_ptr() == action_tape.untyped_storage().data_ptr()), None)
            if hit is None:
                hit = (key, action_tape, self.outs, self.capture_rollout(action_tape, None, True, True))
                cache.append(hit)
                if len(cache) > 8:
                    cache.pop(0)
            hit[3].replay()
            return
        for t in range(T):
            if ring is not None:
                self.use_outputs(ring.sets[t])
            self.step(action_tape[t], auto_reset=True)

    def poison_lds(self):
        """Test hook: fill every CU's LDS with 0xFF bytes (a kernel that reads an LDS table before writing it then fails
        at once instead of seeing its own values of the previous launch)."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fmarl_poison_lds(self.handle, self._stream()), 'fmarl_poison_lds')

    # ------------------------------------------------------------------ measurement
    def profile_enable(self, capacity):
        """Record a hipEvent pair around each of the next ``capacity`` step-kernel launches (0 = off)."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fmarl_profile_enable(self.handle, int(capacity)), 'fmarl_profile_enable')
        self._prof_cap = int(capacity)

    def profile_read(self, with_steps=False):
        """Per-launch step-kernel durations [ms] since profile_enable / the last read (stream must be idle);
        ``with_steps=True``: (durations, env steps each launch covered -- 1 for a step, the run length for a span launch)."""
        buf = (C.c_float * max(self._prof_cap, 1))()
        steps = (C.c_int * max(self._prof_cap, 1))()
        cnt = C.c_int()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fmarl_profile_read(self.handle, buf, steps, self._prof_cap, C.byref(cnt)), 'fmarl_profile_read')
        ms = [buf[i] for i in range(cnt.value)]
        return (ms, [steps[i] for i in range(cnt.value)]) if with_steps else ms

    def launch_geometry(self):
        """(workgroups, threads per workgroup, LDS bytes per workgroup, envs per workgroup) of this engine's step kernels."""
        g = (C.c_int64 * 4)()
        _lib.check(self.lib.fmarl_launch_geometry(self.handle, g), 'fmarl_launch_geometry')
        return tuple(int(x) for x in g)

    def launch_counts(self):
        """(step launches, of which folded episode ends, steps followed by separate auto-reset launches, stagings) so far."""
        buf = (C.c_int64 * 4)()
        _lib.check(self.lib.fmarl_launch_counts(self.handle, buf), 'fmarl_launch_counts')
        return tuple(int(v) for v in buf)

    def close(self):
        if getattr(self, 'handle', None) is not None and self.handle.value:
            with torch.cuda.device(self.device):
                self.lib.fmarl_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _SpreadBlock(object):
    """Device memory of ``fmarl_ring_alloc`` -- time slots whose physical memory is interleaved piece by piece -- presented through
    ``__cuda_array_interface__``; tensors made from it (``torch.as_tensor``) keep it alive, the memory is unmapped with the last one."""

    def __init__(self, lib, device, shape, slot_bytes, slots):
        self._cookie = None
        base, cookie = C.c_void_p(), C.c_void_p()
        with torch.cuda.device(device):
            rc = lib.fmarl_ring_alloc(int(slot_bytes), int(slots), 0, C.byref(base), C.byref(cookie))
        if rc:
            raise MemoryError(lib.fmarl_last_error().decode())
        self._lib, self._cookie, self._device, self.nbytes = lib, cookie, device, int(slot_bytes) * int(slots)
        self.__cuda_array_interface__ = dict(shape=tuple(int(v) for v in shape), typestr='<f4', data=(int(base.value), False), version=2)

    def __del__(self):
        cookie, self._cookie = self._cookie, None
        if cookie is None or not cookie.value:
            return
        try:   # (fmarl_ring_free waits for the device itself before it unmaps)
            if self._lib.fmarl_ring_free(cookie):
                raise RuntimeError(self._lib.fmarl_last_error().decode())
        except Exception as e:
            import logging
            logging.getLogger('fair_marl_amd').warning('fmarl_ring_free failed, %d bytes of device memory stay mapped: %s', self.nbytes, e)


def alloc_time_slots(lib, device, shape, spread=None, zero=False, single=False):
    """A float32 device tensor of ``shape`` = (T, ...) for T time slots: virtually contiguous, and -- ``spread`` True, or None and a slot
    is 64 MiB or more -- with its physical memory interleaved piece by piece over the whole array (``fmarl_ring_alloc``; see
    ``OutputRing``).  Falls back to a plain allocation where the slot size has no suitable divisor or the device has no virtual
    memory management (``spread=True`` raises instead).  ``single``: also for T = 1 (one array: nothing to interleave, but an array made
    of ``hipMemCreate`` pieces takes a store stream faster than one plain allocation of the same size).  -> (tensor, interleaved?)"""
    T, slot_bytes = int(shape[0]), 4 * int(np.prod(shape[1:]))
    if spread is None and os.environ.get('FMARL_RING_SPREAD') == '0':   # (measurement aid: plain allocations)
        spread = False
    if (T >= 2 or single) and (spread or (spread is None and slot_bytes >= (64 << 20))):
        try:
            # (an array allocated after an earlier one of the process was freed has been filled and read back by kernels inside
            # fmarl_ring_alloc -- include/fmarl.h; one that failed raised MemoryError above and the allocation below is a plain one)
            t = torch.as_tensor(_SpreadBlock(lib, device, shape, slot_bytes, T), device=device)
            if zero:
                t.zero_()
            return t, True
        except MemoryError:
            if spread:
                raise
    return (torch.zeros if zero else torch.empty)(*shape, dtype=torch.float32, device=device), False


class OutputRing(object):
    """``slots`` time slots of step outputs, laid out (slots, n, ...) like the reference's rollout storage
    (onpolicy/utils/graph_buffer.py:84-110): slot t is an output set of the engine (``sets[t]``), and ``strides`` are the
    per-step element strides ``RolloutEngine.step_span`` takes, so a run of steps that starts at slot t0 --
    ``engine.use_outputs(ring.sets[t0]); engine.step_span(tape, strides=ring.strides)`` -- leaves step t in slot t0 + t.
    Every step of a rollout has a slot of its own: its outputs exist afterwards and every byte is written once per pass over
    the ring.  Nothing but the time slots (no masks, no policy arrays: ``DeviceRolloutBuffer`` is the runner's buffer).
    At BASELINE config 3 a slot is 8.3 GB: an episode of 25 slots takes 205 of the 288 GB.

    ``like``: another ring of the same shape (same n_envs, config, slots) whose ARRAYS are taken over instead of allocating new
    ones -- a second engine over the same memory.  Worth it for large rings: the first large allocation of a process comes out
    of pristine device memory; one made after it has been freed is, on some boxes, 15 % slower to stream into
    (profiles/r4_notes.md), so allocate the big buffers first and keep them.

    ``spread`` (default: on for node_obs / adj arrays whose slots are 64 MiB and more): the slots stay virtually contiguous --
    ``node_obs[t]`` is an ordinary contiguous tensor -- but their physical memory is interleaved in pieces over the whole array
    (``fmarl_ring_alloc``): MI355X writes ONE 8 GB region at 5.7-6.0 TB/s and the same bytes spread over 160 GB at 6.8-7.1, so a
    launch that fills a single slot (``step``: a policy in the loop) runs at the rate a whole rollout gets.

    ``env_range`` = (first env, count) together with ``like``: this ring is the time slots of a SUB-BATCH of the other ring's envs
    (``PipelinedRollout.new_rings(like=...)``) -- every array is the other ring's, cut along the env axis (slot t of the sub-batch
    is a contiguous block inside slot t of the whole batch; the step-to-step stride is the whole batch's), except the info
    planes, whose (14, n, N) layout puts the env axis second: those are this ring's own."""

    def __init__(self, engine, slots, like=None, spread=None, env_range=None):
        eng, cfg = engine, engine.cfg
        n, N, E, D, F = eng.n_envs, cfg.N, cfg.E, cfg.obs_dim, cfg.node_feat
        self.engine, self.slots = eng, int(slots)
        T = self.slots
        taken = iter(())
        if like is not None:
            taken = iter([like.obs, like.reward, like.done, like.node_obs, like.adj_env, like.info_planes, like.edge_nnz, like.graph_record])

        self.spread = []   # names of the arrays whose physical memory is interleaved

        if env_range is not None and (like is None or env_range[1] != eng.n_envs):
            raise ValueError('OutputRing(env_range=...): needs `like`, and a range of the engine\'s %d envs' % eng.n_envs)

        def mk(*shape, dtype=torch.float32, name=None, own=False):
            old = next(taken, None)
            if own and env_range is not None:
                return torch.empty(*shape, dtype=dtype, device=eng.device)
            if like is not None:
                if old is not None and env_range is not None:
                    old = old[:, env_range[0]:env_range[0] + env_range[1]]
                if old is None or tuple(old.shape) != tuple(shape) or old.dtype != dtype or old.device != eng.device:
                    raise ValueError('OutputRing(like=...): the other ring has no %s array of shape %s' % (dtype, (shape,)))
                if name in getattr(like, 'spread', ()):
                    self.spread.append(name)
                return old
            if name and dtype == torch.float32:
                t, interleaved = alloc_time_slots(eng.lib, eng.device, shape, spread)
                if interleaved:
                    self.spread.append(name)
                return t
            return torch.empty(*shape, dtype=dtype, device=eng.device)
        def skip():   # an array this ring does not have: step over the other ring's
            next(taken, None)
            return None
        with torch.cuda.device(eng.device):
            self.obs = mk(T, n, N, D)
            self.reward = mk(T, n, N)
            self.done = mk(T, n, N, dtype=torch.uint8)
            self.node_obs = mk(T, n, N, E, F, name='node_obs') if eng.emit_graph else skip()
            self.adj_env = mk(T, n, E, E, name='adj') if eng.emit_graph else skip()
            self.info_planes = mk(T, _lib.INFO_WIDTH, n, N, own=True) if eng.emit_info else skip()
            self.edge_nnz = mk(T, n, dtype=torch.int32) if eng.count_edges else skip()
            self.graph_record = mk(T, n, N, eng.step_record_words, dtype=torch.int32) if eng.emit_graph_record else skip()
        pick = lambda a, t: a[t] if a is not None else None  # noqa: E731
        self.sets = [eng.new_output_set(obs=self.obs[t], reward=self.reward[t], done=self.done[t], node_obs=pick(self.node_obs, t),
                                        adj_env=pick(self.adj_env, t), info_planes=pick(self.info_planes, t),
                                        edge_nnz=pick(self.edge_nnz, t), graph_record=pick(self.graph_record, t)) for t in range(T)]
        per = lambda a: int(a.stride(0)) if a is not None else 0  # noqa: E731  (elements from a slot to the next: a sub-batch's view keeps the batch's)
        self.strides = dict(obs=per(self.obs), node_obs=per(self.node_obs), adj=per(self.adj_env), reward=per(self.reward), done=per(self.done),
                            info=per(self.info_planes), edge_nnz=per(self.edge_nnz), graph_record=per(self.graph_record))

    @property
    def nbytes(self):
        return sum(a.numel() * a.element_size() for a in (self.obs, self.reward, self.done, self.node_obs, self.adj_env, self.info_planes,
                                                          self.edge_nnz, self.graph_record) if a is not None)


class _LockstepGraph(object):
    """A graph captured with the reset decisions baked from the host's step mirror: valid from one episode phase only."""

    def __init__(self, engine, graph, phase0, steps):
        import weakref
        # (a weak reference: the engine caches these objects -- RolloutEngine.rollout -- and a reference cycle would leave the
        # graph's destruction to the cyclic collector, which may run in the middle of somebody else's stream capture)
        self._engine, self.graph, self.phase0, self.steps = weakref.ref(engine), graph, phase0, steps

    @property
    def engine(self):
        return self._engine()

    def replay(self):
        eng = self.engine
        if eng.phase != self.phase0:
            raise RuntimeError('graph captured at episode phase %d, the engine is at %d' % (self.phase0, eng.phase))
        self.graph.replay()
        _lib.check(eng.lib.fmarl_set_phase(eng.handle, (self.phase0 + self.steps) % eng.cfg.episode_length), 'fmarl_set_phase')
