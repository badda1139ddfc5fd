#!/usr/bin/env python
"""Throughput of the rollout hot path: random-action rollouts of navigation_graph on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one env step of EVERY env of every rank (one pass of the hot path over one batch):
action decode -> World.step physics -> obs / reward / info / node_obs / adj emission, plus the
auto-reset (placement + fair assignment + reset observation) whenever an episode ends (every
25th step).  Inputs (the random action tape) and all outputs stay resident in HBM; every step's
outputs go to a time slot of their own (--slots ring: an episode-long ring of (T, n, ...) arrays, the
layout of the reference's rollout storage), so the trajectory exists when the rollout is over and
every byte of it is written once per pass.

Metric (BASELINE.json): agent-steps/s = total envs x agents x K / wall seconds, where wall is
the max over ranks of the time of exactly K steps bracketed by barrier + device sync.
With N > 1 each rank owns 65 536 envs (weak scaling) and every step's trajectory record
(obs, reward, done) is gathered to rank 0 over RCCL inside the timed region.

Rank 0 prints ONE compact JSON line on stdout (<= 6 000 bytes: compact_line) and writes the full record -- the prose, the tables,
every secondary entry -- to bench_detail.json and stderr; see DESIGN.md "Measurement" for how roofline / cpu_baseline are built.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import fair_marl_amd as fm  # noqa: E402
from fair_marl_amd.sharding import SpanGather, StepRecord, TrajectoryGather  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

CONFIGS = {   # workload: a format string, filled with the number of envs the run really uses (--n-envs)
    # BASELINE.json configs[2] (and configs[4] per GPU): the configuration the target is quoted on
    'cfg3': dict(workload='navigation_graph, 32 agents + 8 obstacles (E=72), %d envs per GPU',
                 env=dict(num_agents=32, num_landmarks=32, num_obstacles=8), n_envs=65536, cpu_envs=32, cpu_episodes=12),
    # BASELINE.json configs[3]
    'cfg4': dict(workload='fair_graph_formation, 10 agents + 1 landmark + 3 obstacles + 2 walls (E=16), %d envs per GPU',
                 env=dict(scenario_name='fair_graph_formation', num_agents=10, num_landmarks=1, num_obstacles=3),
                 n_envs=65536, cpu_envs=16, cpu_episodes=12),
    # the shipped FA+FR weights' configuration (model_weights/FA+FR/config.yaml)
    'fnav': dict(workload='nav_fairassign_fairrew_formation_graph, 3 agents + 3 obstacles (E=9), %d envs per GPU',
                 env=dict(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=3, num_landmarks=3,
                          num_obstacles=3, goal_rew=30.0, collision_rew=30.0), n_envs=65536, cpu_envs=32, cpu_episodes=4),
    # the same scenario at the size BASELINE.md section 2 times the reference at (N = 10: 855 agent-steps/s on one core): 64 x N > 192 lanes,
    # so fairnav_span_kernel<256>, and the per-step re-assignment grows with N (nav_fairassign_fairrew_formation_graph.py:704-721)
    'fnav10': dict(workload='nav_fairassign_fairrew_formation_graph, 10 agents + 10 goals + 3 obstacles (E=23), %d envs per GPU',
                   env=dict(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=10, num_landmarks=10,
                            num_obstacles=3, goal_rew=30.0, collision_rew=30.0), n_envs=65536, cpu_envs=16, cpu_episodes=3),
    # the reference's own experiment scale (10 agents): odd E, so the generic (row-per-lane) emission path
    'n10': dict(workload='navigation_graph, 10 agents + 3 obstacles (E=23), %d envs per GPU',
                env=dict(num_agents=10, num_landmarks=10, num_obstacles=3), n_envs=65536, cpu_envs=64, cpu_episodes=3),
    # BASELINE.json configs[1]
    'cfg2': dict(workload='navigation_graph, 3 agents + 3 obstacles (E=9), %d envs per GPU',
                 env=dict(num_agents=3, num_landmarks=3, num_obstacles=3), n_envs=4096, cpu_envs=512, cpu_episodes=20),
}
KERNEL_NAMES = {'fair_graph_formation': 'formation_kernel<true>', 'nav_fairassign_fairrew_formation_graph': 'fairnav_kernel<true>'}
# every BASELINE config that fits one GPU besides the headline one, timed after the headline region (same process, fresh
# engines) and reported under `secondary`: (config, launch mode)
SECONDARY = (('cfg3', 'eager'), ('cfg3', 'pipeline2'), ('cfg3', 'span'), ('cfg3', 'span5'), ('cfg3', 'span-same'), ('cfg2', 'span'), ('cfg2', 'graph'), ('cfg2', 'eager'),
             ('cfg4', 'eager'), ('cfg4', 'span'), ('cfg4', 'pipeline2'), ('fnav', 'eager'), ('fnav', 'span'), ('fnav', 'pipeline2'), ('fnav', 'steady'), ('fnav', 'steady-span'),
             ('fnav10', 'eager'), ('fnav10', 'span'),
             ('n10', 'eager'), ('n10', 'span'), ('n10', 'pipeline2span'))
# nav_fairassign_fairrew_formation_graph where a training run is: a threshold at which goals are reached, so that episodes end
# env by env, and enough untimed steps for the envs' episode phases to be uniform (mode 'steady'; round 5: tools/archive/fnav_steady.py, in the history up to 28d23e2)
STEADY = dict(min_dist_thresh=0.5, pre_steps=600)


# The reference's OWN CPU path (GraphSubprocVecEnv, one process per env), timed in the build container where the reference
# can be imported (it cannot travel to the GPU box): quoted beside cpu_baseline, never a target.
_REF_HW = '8 Xeon 2.1 GHz cores, build container (Gurobi replaced by the pure-Python lexifair solver)'
REFERENCE_CPU = {
    'cfg3': dict(value=1662.0, unit='agent-steps/s', cores=8, hardware=_REF_HW, source='BASELINE.md section 2: GraphSubprocVecEnv x 8, N=32, O=8'),
    'cfg2': dict(value=5142.0, unit='agent-steps/s', cores=8, hardware=_REF_HW, source='BASELINE.md section 2: GraphSubprocVecEnv x 8, N=3'),
    'cfg4': dict(value=1716.0, unit='agent-steps/s', cores=8, hardware=_REF_HW, source='BASELINE.md section 2: fair_graph_formation, GraphSubprocVecEnv x 8, N=10'),
}


def algorithmic_bytes(cfg, emit=True):
    """bytes per agent-step, B = 4 [A + S + D + R + F E + E^2/N + C/N]."""
    N, E = cfg.N, cfg.E
    C = 2 * (cfg.num_landmarks + cfg.num_obstacles) + 6 * cfg.num_walls
    if cfg.scenario_name == 'fair_graph_formation':
        C += 6 * N   # slot words read + written per env (SURVEY section 8 d, cfg 4 row)
    A, S, R = 1, 2 * 12, 3
    words = A + S + R
    if emit:
        words += cfg.obs_dim + cfg.node_feat * E + E * E / N + C / N
    return 4.0 * words


def _scatter_order(chunks):
    import math
    o = int(chunks * 0.6180339887) | 1
    while math.gcd(o, chunks) != 1:
        o += 2
    return o


def store_ceiling(device, step_bytes, steps=1, launches=2, dst=None, earlier=None):
    """The box's write ceiling for a launch that writes `steps` steps' bytes: the best pure 16-byte store stream over that
    byte count (fmarl_store_stream: kernels that do nothing but write, in the shapes the emission writes in -- a workgroup's
    contiguous chunk, a wave's contiguous quarter of one -- in dispatch order and scattered over the buffer).  No step kernel
    can be faster than its own store stream, so kernel time per step / this figure <= 1 by construction -- unlike the
    emission-only launch of rounds 2-3 (`emission_only_ms`), which a span could beat.  Capped at 4 steps' bytes (33 GB at
    cfg 3): a longer stream only amortises the same head and tail further (24 steps: 1.151 ms per step, one step: 1.163).
    A ceiling is the FASTEST the box writes: every launch is timed on its own and the best one counts, and `earlier` (the same
    measurement taken before the run's warm-up) is merged in -- boxes slow down under sustained store load (a 300-step region ran
    at 1.51 ms per step on a box whose 20-step region ran at 1.23, profiles/r4_notes.md), and a ceiling taken only after the
    timed region would be beaten by a short region.
    `dst`: the tensor to write into -- the run's own output buffer (the node_obs slots of the ring, or the engine's node_obs): how
    fast a store stream runs depends on the physical pages a buffer got (the same stream measured 5.6 and 7.1 TB/s in two
    processes of one box, profiles/r4_notes.md), so the ceiling is taken on the pages the kernel itself wrote; a buffer smaller
    than the byte count is written whole and the time scaled by bytes.  Without `dst` a scratch buffer is allocated.
    Returns dict(ms_per_step, TBps, shape, streams={shape: ms per step})."""
    import ctypes as C
    from fair_marl_amd import _lib
    lib = _lib.load()
    step_bytes = int(step_bytes) // 16 * 16
    k = max(1, min(int(round(steps)), 4))
    if dst is not None:
        # a span writes all its time slots at once (its workgroups drift apart in time): the comparable stream covers the whole
        # buffer -- the same footprint over the HBM stacks -- not only its first slots (a 33 GB corner of the ring took a stream at
        # 5.97 TB/s on a box whose span kernel wrote the 205 GB ring at 6.88: profiles/r4_notes.md)
        cap = dst.numel() * dst.element_size() // 16 * 16
        nbytes = min((24 if steps > 1 else 1) * step_bytes, cap)
        if nbytes < (1 << 16):
            return None
        buf = dst.view(-1).view(torch.uint8)[:nbytes]
    else:
        free, _ = torch.cuda.mem_get_info(device)
        while k > 1 and k * step_bytes > free * 0.8:
            k -= 1
        nbytes = k * step_bytes
        if nbytes < (1 << 16) or nbytes > free * 0.8:
            return None
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    out = {}
    for shape, chunk, scat in ((1, 1 << 20, False), (2, 1 << 20, True), (2, 1 << 16, True)):
        chunk = min(chunk, max(4096, nbytes // 64 // 16 * 16))
        chunks = (nbytes // 16 + chunk // 16 - 1) // (chunk // 16)
        order = _scatter_order(chunks) if scat and chunks >= 64 else 1
        call = lambda: _lib.check(lib.fmarl_store_stream(buf.data_ptr(), nbytes, shape, chunk, order, 0, st), 'fmarl_store_stream')  # noqa: E731
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
        call()
        ev[0].record()
        for j in range(launches):
            call()
            ev[j + 1].record()
        ev[-1].synchronize()
        best_ms = min(ev[j].elapsed_time(ev[j + 1]) for j in range(launches))
        out['shape %d, %d KB chunks, %s' % (shape, chunk >> 10, 'scattered' if order > 1 else 'dispatch order')] = best_ms * step_bytes / nbytes
    del buf
    if dst is None:
        torch.cuda.empty_cache()
    if earlier:
        for key, ms in earlier['streams'].items():
            out[key] = min(out.get(key, ms), ms)
    best = min(out, key=out.get)
    return dict(ms_per_step=out[best], TBps=step_bytes / out[best] / 1e9, shape=best, bytes_per_launch=nbytes,
                destination='the run\'s own node_obs buffer' if dst is not None else 'a scratch buffer',
                basis='best single launch per stream, before the warm-up and after the timed region' if earlier else 'best single launch per stream', streams=out)


def emission_only_ms(eng, launches=10):
    """What the emission alone costs on THIS box and THESE buffers: the pure emission kernel (fmarl_rebuild_graph: writes
    node_obs + adj of every env from obs + the episode record, touches no env state) timed on the engine's current output
    buffers.  One launch per step's bytes incl. its head and tail -- a reference point for the one-launch-per-step mode, not a
    ceiling (that is `store_ceiling`).  navigation_graph only (the two formation scenarios rebuild from a per-step record the
    engine does not write by default); call it after the timed region, it overwrites node_obs / adj."""
    if eng.cfg.scenario_name != 'navigation_graph' or eng.node_obs is None:
        return None
    rec = eng.pack_episode()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        eng.rebuild_graph(eng.obs, rec, node_obs=eng.node_obs, adj_env=eng.adj_env)
    e0.record()
    for _ in range(launches):
        eng.rebuild_graph(eng.obs, rec, node_obs=eng.node_obs, adj_env=eng.adj_env)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / launches


def moved_bytes(cfg, agents, steps):
    """Bytes a span launch of `steps` steps really moves by the formula's own terms: everything but the state per step, the state
    (S = 2 x 12 words) and the static entity words once per launch -- the span kernels keep both on the chip between the steps
    (the per-step formula counts them every step)."""
    per_step = algorithmic_bytes(cfg)
    C = 2 * (cfg.num_landmarks + cfg.num_obstacles) + 6 * cfg.num_walls
    once = 4.0 * (2 * 12 + C / cfg.N)   # (all three scenarios' spans since round 5: nav_fairassign_fairrew_formation_graph carries its state too)
    return agents * ((per_step - once) * steps + once)


def launch_bytes(cfg, agents, counts0, counts1):
    """Mean algorithmic bytes per step-kernel launch between two fmarl_launch_counts readings: a launch emits the full outputs
    unless it is an episode-ending step whose reset observation is written by separate reset launches (counts[2])."""
    launches = counts1[0] - counts0[0]
    quiet = counts1[2] - counts0[2] if cfg.scenario_name != 'nav_fairassign_fairrew_formation_graph' else 0
    return agents * (algorithmic_bytes(cfg) * (launches - quiet) + algorithmic_bytes(cfg, emit=False) * quiet) / max(1, launches)


# envs per workgroup of span launches (0 / absent = the library's choice).  Spans into time slots (the default) run fastest at the
# library's choice; spans that rewrite ONE output set every step (--slots same, round 3's mode) at one env per wave for cfg 3
# (profiles/r4_span_slots.md: 1.24 ms per step at 8 envs per workgroup into slots, 1.33 at 4 / 1.41 at 8 into the same set)
SPAN_EPB = {}
SAME_SLOT_EPB = {'cfg3': 4}
EAGER_EPB = {}   # one launch per step: the library's choice everywhere (tools/ring_epb.py <config> <list> eager)
# steps per span launch when the records of a run travel to a learner rank (N > 1): a run's records can only leave when its launch
# has ended, so the LAST run's gather of a timed region is exposed in full -- short runs keep it short, long runs save launches
GATHER_SPAN_STEPS = 5   # (and the last runs of a region halve down to single steps: run_spans)
# N > 1 without --span-steps: the run length is CHOSEN during the warm-up -- the first 8-GPU run cannot be repeated with another constant.
# For every candidate a few runs with the exchange on; kept: the fastest length (time per step, waits included, max over ranks) among
# those whose gather waits stall the rollout by less than SPAN_TUNE_STALL of its time (pick_span_length).
SPAN_TUNE_CANDIDATES = (3, 5, 8, 12)
SPAN_TUNE_RUNS = 4
SPAN_TUNE_STALL = 0.02
# the secondary configs other than the headline's run in child processes of their own, started before this process touches the GPU,
# smallest footprint first: what an entry measures then does not depend on what ran before it (profiles/r4_notes.md section 18: behind
# the headline's 205 GB ring `n10 eager` read 0.206-0.273 ms per step, in a fresh process 0.2112-0.2119)
SECONDARY_CHILD_ORDER = ('cfg2', 'fnav', 'cfg4', 'fnav10', 'n10')


def _small_batch(cfg, eng):
    """navigation_graph batches whose workgroups hold all their agents in one wave and emit rows of generic shape run the small-batch
    kernels (fmarl_step.hip step_body SMALL: step_small_kernel / step_span_small_kernel) -- BASELINE config 2."""
    return cfg.scenario_name == 'navigation_graph' and eng.envs_per_workgroup * cfg.N <= 64 and (cfg.E * cfg.node_feat) % 4 != 0


def _use_ring(cfg, n_envs, slots, device):
    """Time slots for every step (an episode-long OutputRing) unless the caller asked for one output set or the ring would not fit."""
    if slots != 'ring':
        return False
    per_slot = n_envs * cfg.N * 4.0 * (cfg.obs_dim + 1 + 0.25 + 14 + cfg.node_feat * cfg.E + cfg.E * cfg.E / cfg.N)
    free, _ = torch.cuda.mem_get_info(device)
    return per_slot * cfg.episode_length < free * 0.85


def secondary_line(name, mode, device, steps=300, warmup=50, slots='ring', arena=None):
    """One more BASELINE config on this GPU: fresh engine, `warmup` untimed steps (whole episodes), `steps` timed steps (whole
    episodes), synchronised on both sides, all through RolloutEngine.rollout(tape, mode, ring):
      'eager' = one fmarl_step call per step (what a policy in the loop gets);
      'span'  = fmarl_step_span: the steps between episode ends as ONE launch in which every workgroup walks its own envs through
                time, the step that ends the episode as a launch of its own (random-action / scripted rollouts: actions known
                ahead, the metric of BASELINE.json);
      'span<S>' = the same in runs of at most S steps (the launch pattern of an N > 1 run, whose records leave run by run);
      'span-same' = spans that rewrite ONE output set every step (stride 0: round 3's headline mode, kept for comparison);
      'graph' = one hipGraph replay per episode (launch-bound batches; one output set);
      'steady' (nav_fairassign_fairrew_formation_graph) = one launch per step with episodes ending at all phases (STEADY),
      'steady-span' = the same regime as spans (that scenario's span: the whole tape one launch, episode ends inside);
      'pipeline<k>' = k sub-batches on k streams.
    Every step writes its own time slot of an episode-long ring (`slots`: 'ring') except where the mode says otherwise; `arena` = a
    ring of the same shape whose arrays are taken over (the headline's: a large allocation made after another one was freed is, on
    some boxes, 15 % slower to stream into than the first one of the process -- profiles/r4_notes.md).
    kernel_avg_ms is per STEP in every mode (a span launch's duration divided by its steps)."""
    spec = CONFIGS[name]
    env_kw = dict(spec['env'])
    pre_steps = 0
    if mode.startswith('steady'):
        env_kw['min_dist_thresh'] = STEADY['min_dist_thresh']
        pre_steps = STEADY['pre_steps']
    cfg = fm.EnvConfig(**env_kw)
    n = spec['n_envs']
    ep = cfg.episode_length
    steps, warmup = max(ep, steps // ep * ep), (warmup + ep - 1) // ep * ep
    if mode.startswith('pipeline'):
        spans = mode.endswith('span')
        return secondary_pipeline(name, int(mode[len('pipeline'):].replace('span', '')), device, steps, warmup, spans, slots, arena=arena)
    same = mode == 'span-same'
    run_len = int(mode[4:]) if mode.startswith('span') and mode[4:].isdigit() else 0
    rmode = 'span' if mode.startswith('span') or mode == 'steady-span' else ('eager' if mode == 'steady' else mode)
    epb = (SAME_SLOT_EPB if same else SPAN_EPB).get(name, 0) if rmode == 'span' else EAGER_EPB.get(name, 0)
    eng = fm.RolloutEngine(cfg, n, device=device, seed=1, envs_per_workgroup=epb, tune_placement=0)
    ring = None
    if not same and rmode != 'graph' and (arena is not None or _use_ring(cfg, n, slots, device)):
        ring = fm.OutputRing(eng, ep, like=arena)
    g = torch.Generator(device=device)
    g.manual_seed(2000)
    tape = torch.randint(0, 5, (ep, n, cfg.N), device=device, generator=g, dtype=torch.int32)
    eng.reset()
    agents_bytes = algorithmic_bytes(cfg) * n * cfg.N
    # (the write ceiling is read where the run writes time slots: a stream into ONE output set's 6.6 GB covers a smaller footprint than
    # the kernel's own node_obs + adj and is no ceiling for it)
    ceil_dst = ring.node_obs if ring is not None else None
    ceil0 = store_ceiling(device, agents_bytes, (run_len or ep - 1) if rmode == 'span' else 1, dst=ceil_dst) if ceil_dst is not None else None

    def episode(m):
        if run_len and m == 'span':   # runs of at most run_len steps (fnmarl_step_span splits at the episode end by itself)
            for off in range(0, ep, run_len):
                k = min(run_len, ep - off)
                if ring is not None:
                    eng.use_outputs(ring.sets[off])
                eng.step_span(tape[off:off + k], strides=ring.strides if ring is not None else None)
        else:
            eng.rollout(tape, mode=m, ring=ring)
    timed = rmode if rmode != 'graph' else 'eager'   # hipEvents cannot live inside a graph: kernel time from an eager pass
    for _ in range(max(1, warmup // ep) + pre_steps // ep):
        episode(timed)
    torch.cuda.synchronize(device)
    # one launch per step: an event pair around every launch sits BETWEEN the launches (6-7 us per step at 65 536 x 3: a tenth of the
    # step) -- the wall time comes from a pass without them, the kernel times from a second pass of the same steps
    # (the same for the spans of a launch-bound batch: four event records per episode of 150 us)
    two_pass = rmode == 'eager' or (rmode == 'span' and n * cfg.N < fm.RolloutEngine.GRAPH_BELOW_AGENTS)
    eng.profile_enable(0 if two_pass else steps)
    c0 = eng.launch_counts()
    if rmode == 'graph':
        for _ in range(2):
            eng.rollout(tape, mode='eager')
        torch.cuda.synchronize(device)
        kernel_ms, c1 = eng.profile_read(), eng.launch_counts()
        eng.profile_enable(0)
        eng.rollout(tape, mode='graph')          # capture + first replay
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps // ep):
        episode(rmode)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if two_pass:
        eng.profile_enable(steps)
        c0 = eng.launch_counts()
        for _ in range(steps // ep):
            episode(rmode)
        torch.cuda.synchronize(device)
    if rmode != 'graph':
        kernel_ms, c1 = eng.profile_read(), eng.launch_counts()
    k_step = float(np.sum(kernel_ms)) / (c1[0] - c0[0])       # step-kernel time per step (a span launch covers many)
    per_step = launch_bytes(cfg, n * cfg.N, c0, c1)
    folded = c1[1] - c0[1] > 0
    small = _small_batch(cfg, eng)
    kern = KERNEL_NAMES.get(cfg.scenario_name, ('step_small_kernel' if small else 'step_kernel') + (' / step_end_kernel' if folded else ''))
    if rmode == 'span':
        kern = {'fair_graph_formation': 'formation_span_kernel + formation_kernel<true>',
                'nav_fairassign_fairrew_formation_graph': 'fairnav_span_kernel'}.get(cfg.scenario_name, ('step_span_small_kernel' if small else 'step_span_kernel') + ' + step_end_kernel')
    span_text = 'fmarl_step_span: one launch per run of steps between episode ends%s (%d envs per workgroup), the episode-ending step a launch of its own' \
                % (' and at most %d steps' % run_len if run_len else '', eng.envs_per_workgroup)
    if cfg.scenario_name == 'nav_fairassign_fairrew_formation_graph':
        span_text = ('fmarl_step_span: the %d steps of the tape as ONE launch (%d envs per workgroup); episodes end env by env and the step '
                     'resets them itself; the state stays in registers between the steps (three waves per workgroup)' % (ep, eng.envs_per_workgroup))
    steps_per_launch = (c1[0] - c0[0]) / max(1, len(kernel_ms))
    if rmode == 'span' and steps_per_launch <= 1.0:
        # fmarl_step_span launches per step where a span form does not pay (nav_fairassign_fairrew_formation_graph beyond three agents:
        # profiles/r6_fnav_spans_by_n.txt): say which kernel ran
        kern = KERNEL_NAMES.get(cfg.scenario_name, kern)
        span_text += '; at this shape the library launches per step inside fmarl_step_span'
    epw = eng.envs_per_workgroup
    ring_bytes = ring.nbytes if ring is not None else 0
    # (on the pages the kernel wrote: the ring's node_obs slots, else the engine's node_obs)
    ceil = store_ceiling(device, agents_bytes, steps_per_launch if rmode == 'span' else 1, dst=ceil_dst, earlier=ceil0) if ceil_dst is not None else None
    eng.close()
    del eng, tape, ring, ceil_dst
    torch.cuda.empty_cache()
    out = dict(config=name, mode=mode, workload=spec['workload'] % n,
               launch={'eager': 'one fmarl_step call per step', 'span': span_text,
                       'graph': 'one hipGraph replay per episode, the staged reset a forked branch of the graph (kernel_avg_ms from an eager pass)'}[rmode],
               slots=('every step its own time slot of an episode-long ring (%.1f GB)' % (ring_bytes / 1e9) if ring_bytes
                      else 'one output set, rewritten every step'),
               value=n * cfg.N * steps / elapsed, unit='agent-steps/s', steps=steps, warmup=warmup + pre_steps // ep * ep, ms_per_step=elapsed / steps * 1e3,
               kernel=kern, kernel_avg_ms=k_step, kernel_launches=len(kernel_ms), envs_per_workgroup=epw,
               frac=per_step / (k_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
               algorithmic_bytes_per_step=per_step, store_ceiling_ms=ceil['ms_per_step'] if ceil else None,
               store_ceiling_shape=ceil['shape'] if ceil else None,
               frac_of_box_ceiling=(ceil['ms_per_step'] / k_step if ceil else None))
    if n * cfg.N < fm.RolloutEngine.GRAPH_BELOW_AGENTS:
        out['bound'] = ('launch / latency: the batch cannot fill the chip (%d waves on 256 CUs), a step is one wave\'s dependent float64 chain; '
                        'an HBM roofline does not apply -- `frac` is reported for completeness' % ((n * cfg.N + 63) // 64))
    if two_pass:
        out['timing'] = ('ms_per_step: a pass without per-launch events; kernel_avg_ms: a second pass of the same steps with a hipEvent pair '
                         'around every launch (the pairs sit between the launches and delay their dispatch: kernel_avg_ms can exceed ms_per_step)')
    if mode.startswith('steady'):
        out['regime'] = 'min_dist_thresh %.2f, %d untimed steps first: episodes end env by env at all phases' % (STEADY['min_dist_thresh'], pre_steps)
    return out


def secondary_pipeline(name, k, device, steps, warmup, spans=False, slots='ring', arena=None):
    """The same envs as k sub-batches on k streams (fair_marl_amd.PipelinedRollout; bit-identical results): the tail of one
    sub-batch's launch overlaps the head of another's.  What a random-action rollout -- actions known ahead -- or an
    alternating sampler gets out of the chip for the compute-heavy scenarios; with ``spans`` every sub-batch runs its steps as
    spans (PipelinedRollout.rollout).  `frac` is the whole job's algorithmic bytes per step over the time per step (a launch
    that shares the chip is longer than it would be alone)."""
    spec = CONFIGS[name]
    cfg = fm.EnvConfig(**spec['env'])
    n, ep = spec['n_envs'], cfg.episode_length
    pipe = fm.PipelinedRollout(cfg, n, k=k, device=device, seed=1, tune_placement=0)
    # (`arena`: the headline's ring -- every sub-batch writes its envs' part of the same (T, n, ...) time slots)
    rings = pipe.new_rings(ep, like=arena) if arena is not None else (pipe.new_rings(ep) if _use_ring(cfg, n, slots, device) else None)
    g = torch.Generator(device=device)
    g.manual_seed(2000)
    tape = torch.randint(0, 5, (ep, n, cfg.N), device=device, generator=g, dtype=torch.int32)
    tapes = pipe.split_tape(tape)
    mode = 'span' if spans else 'eager'
    pipe.reset()
    for _ in range(max(1, warmup // ep)):
        pipe.rollout(tapes, mode=mode, rings=rings)
    pipe.synchronize()
    torch.cuda.synchronize(device)
    for e in pipe.engines:
        e.profile_enable(steps)
    c0 = [e.launch_counts() for e in pipe.engines]
    t0 = time.perf_counter()
    for _ in range(steps // ep):
        pipe.rollout(tapes, mode=mode, rings=rings)
    pipe.synchronize()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    kernel_ms = [v for e in pipe.engines for v in e.profile_read()]
    c1 = [e.launch_counts() for e in pipe.engines]
    per_step_sub = float(np.mean([launch_bytes(cfg, (n // k) * cfg.N, a, b) for a, b in zip(c0, c1)]))
    job = per_step_sub * k / (elapsed / steps) / 1e9
    out = dict(config=name, mode='pipeline%d%s' % (k, 'span' if spans else ''), workload=spec['workload'] % n,
               launch='%d sub-batches of %d envs on their own streams, each %s' % (k, n // k, 'running its steps as spans (fmarl_step_span)'
                                                                                   if spans else 'one fmarl_step call per step'),
               slots=('every step its own time slot of the whole batch\'s episode-long ring (a sub-batch writes its envs\' part of a slot)' if arena is not None else
                      'every step its own time slot of an episode-long ring per sub-batch' if rings is not None else 'one output set per sub-batch, rewritten every step'),
               value=n * cfg.N * steps / elapsed, unit='agent-steps/s', steps=steps, warmup=warmup,
               ms_per_step=elapsed / steps * 1e3, kernel=KERNEL_NAMES.get(cfg.scenario_name, 'step_span_kernel + step_end_kernel' if spans else 'step_kernel / step_end_kernel'),
               kernel_avg_ms=float(np.sum(kernel_ms)) / sum(b[0] - a[0] for a, b in zip(c0, c1)), kernel_launches=len(kernel_ms), frac=job / HBM_PEAK_GBS,
               frac_basis='whole job: algorithmic bytes per step of all envs / time per step (kernel_avg_ms: a sub-batch\'s step kernels per step, while others run)',
               algorithmic_bytes_per_step=per_step_sub * k, store_ceiling_ms=None, frac_of_box_ceiling=None)
    pipe.close()
    del pipe, tape, tapes, rings
    torch.cuda.empty_cache()
    return out


def _cpu_worker(job):
    """One host process = one batch of envs stepped by the oracle (the reference runs one process per env)."""
    env_kw, n_envs, episodes, seed = job
    from oracle import fairnav_oracle as fnv
    from oracle import formation_oracle as fo
    from oracle import nav_oracle as no
    from oracle.philox import PhiloxStream
    cfg = fm.EnvConfig(**env_kw)
    streams = lambda e, ep: PhiloxStream(seed, e, ep)  # noqa: E731
    if cfg.scenario_name == 'nav_fairassign_fairrew_formation_graph':
        ocfg = fnv.Config(**{k: getattr(cfg, k) for k in fnv.Config.__dataclass_fields__})
        env = fnv.OracleFairNavVecEnv(ocfg, n_envs, mode='subproc', streams=streams)
    elif cfg.scenario_name == 'fair_graph_formation':
        ocfg = fo.Config(**{k: getattr(cfg, k) for k in fo.Config.__dataclass_fields__})
        env = fo.OracleFormationVecEnv(ocfg, n_envs, mode='subproc', streams=streams)
    else:
        ocfg = no.Config(**{k: getattr(cfg, k) for k in no.Config.__dataclass_fields__})
        env = no.OracleGraphVecEnv(ocfg, n_envs, mode='subproc', streams=streams)
    env.reset()
    rs = np.random.RandomState(seed)
    steps = episodes * cfg.episode_length
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step(rs.randint(0, 5, size=(n_envs, cfg.N)))
    return n_envs * cfg.N * steps, time.perf_counter() - t0


def cpu_baseline(env_kw, n_envs, episodes, workers):
    """The oracle (NumPy float64 restatement) on a bounded sample of the same workload, one process per host
    core like the reference's SubprocVecEnv (capped at 16).  Must run BEFORE this process touches the GPU: the
    workers are spawned (exec), which is not allowed once HIP is initialised."""
    jobs = [(env_kw, n_envs, episodes, 1 + w) for w in range(workers)]
    t0 = time.perf_counter()
    if workers > 1:
        import multiprocessing as mp
        with mp.get_context('spawn').Pool(workers) as pool:
            res = pool.map_async(_cpu_worker, jobs).get(timeout=240)   # never hang the bench on the baseline
    else:
        res = [_cpu_worker(jobs[0])]
    wall = time.perf_counter() - t0
    units = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    N, ep = res[0][0] // (n_envs * episodes), episodes
    return dict(value=units / busy, unit='agent-steps/s', cores=workers, kind='port',
                sample='%d processes x %d envs x %d episodes incl. auto-resets (%d agent-steps), NumPy f64 oracle, '
                       'slowest worker %.1f s, %.1f s wall incl. process start' % (workers, n_envs, ep, units, busy, wall))


def secondary_child(name):
    """`bench.py --secondary-child NAME`: every `secondary` entry of one config on a fresh process, as one JSON line."""
    device = torch.device('cuda', 0)
    torch.cuda.set_device(device)
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    t0 = time.perf_counter()
    entries = [secondary_line(n, mode, device) for n, mode in SECONDARY if n == name]
    for e in entries:
        e['process'] = 'child process of its own, started before the headline run touched the GPU'
    os.write(result_fd, (json.dumps({'config': name, 'entries': entries, 'wall_s': time.perf_counter() - t0}) + '\n').encode())


def run_secondary_children(headline):
    """The secondary entries of every config but the headline's, each config in a fresh child process (this process must not have
    initialised the GPU yet).  -> ({config: [entries]}, {config: error text}, wall seconds)"""
    import subprocess
    done, failed = {}, {}
    t0 = time.perf_counter()
    for name in SECONDARY_CHILD_ORDER:
        if name == headline or not any(n == name for n, _ in SECONDARY):
            continue
        try:
            # (a child runs all modes of its config: 60 s per entry + 600 untimed steps for the steady ones, on top of 120 s of start-up)
            entries = sum(1 for n, _ in SECONDARY if n == name)
            res = subprocess.run([sys.executable, os.path.abspath(__file__), '--secondary-child', name], capture_output=True, text=True,
                                 timeout=120 + 60 * entries)
            lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
            if res.returncode != 0 or len(lines) != 1:
                raise RuntimeError('exit code %d: %s' % (res.returncode, res.stderr[-400:]))
            done[name] = json.loads(lines[0])['entries']
        except Exception as exc:   # the entries then run in this process after the headline, and say so
            failed[name] = str(exc)[-400:]
            print('bench.py: secondary child for %s failed (%s); its entries run in-process' % (name, exc), file=sys.stderr)
    return done, failed, time.perf_counter() - t0


def span_schedule(first, count, episode_length, span_steps, taper):
    """Run lengths of the steps [first, first + count) as spans: a run ends at an episode end, after `span_steps` steps or with the
    region; `taper` (the records of a run travel to a learner rank): the last stretch of the region halves down to single steps, so
    that the gather nobody can hide -- the last one, behind which the region's closing synchronisation waits -- carries one step's
    records instead of a whole run's."""
    runs, t, end = [], first, first + count
    while t < end:
        k = min(episode_length - t % episode_length, end - t, span_steps)
        if taper and end - t <= span_steps:
            k = min(k, max(1, (end - t + 1) // 2))
        runs.append(k)
        t += k
    return runs


def pick_span_length(table):
    """Among the run lengths whose gather waits (max over ranks; of the compute stream or of the host) stay under 2 % of the time per step:
    the one with the lowest measured time per step, ties to the shorter; if every length stalls: the lowest time per step of all.
    (Long runs save launch heads and tails, short runs keep the exposed gathers small: the measured time per step, waits included,
    is what the run is after -- the stall share only decides who is admissible.)"""
    ok = [r for r in table if r['stall_frac'] < SPAN_TUNE_STALL] or table
    return min(ok, key=lambda r: (r['ms_per_step'], r['span_steps']))


def launch_plan(launch, pipeline, scenario_name, gather, span_steps, episode_length):
    """(launch mode, most steps per span launch) of a run.  The launch mode never depends on the number of GPUs: spans, except for
    nav_fairassign_fairrew_formation_graph, which steps -- its envs end their episodes INSIDE a span, and the record of an episode's
    static entities, which the learner rank needs for every step, can only be packed between launches (its span figure at N = 1 is
    the `secondary` entry (fnav, span)).  What the exchange
    changes is the LENGTH of a run: a run's records can only leave when its launch has ended, so with a gather the runs are
    GATHER_SPAN_STEPS long -- the gather of one run crosses xGMI while the next one computes, and only the last run's gather of
    a timed region is exposed -- instead of reaching to the episode end."""
    if pipeline > 1:
        launch = 'step'
    if launch == 'auto':
        launch = 'step' if scenario_name == 'nav_fairassign_fairrew_formation_graph' else 'span'
    steps = span_steps if span_steps > 0 else (GATHER_SPAN_STEPS if gather else episode_length)
    return launch, max(1, min(steps, episode_length))


# ---- the line the driver parses -------------------------------------------------------------------------------------------------
# stdout carries ONE compact JSON line (<= COMPACT_LIMIT bytes: round 5's 21 KB line was more than the driver's parser took, its record
# came back `parsed: null`).  Scalars and one-word modes only; the prose (what a launch mode is, how a figure was taken), the store
# streams' table, the span-tuning table, the per-rank rows and the full `secondary` entries go to bench_detail.json beside this
# file (--detail PATH) and, as one line behind DETAIL_PREFIX, to stderr.
COMPACT_LIMIT = 6000
DETAIL_PREFIX = 'bench.py detail: '
_TOP_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'n_ranks_seen', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
             'vs_baseline', 'dtype', 'data')
_CONFIG_KEYS = ('workload', 'n_envs_per_gpu', 'n_agents', 'n_entities', 'episode_length', 'auto_resets_timed', 'launch_mode', 'span_steps',
                'slots', 'reset', 'exchange')
_ROOFLINE_KEYS = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'kernel_avg_ms', 'kernel_launches', 'kernel_steps_per_launch',
                  'store_ceiling_ms', 'frac_of_box_ceiling', 'frac_bytes_moved', 'slots')
SECONDARY_FIELDS = ('config', 'mode', 'ms_per_step', 'frac')


def _sig(v, digits=6):
    """Floats to `digits` significant digits (the line is read by people and a parser, neither needs 17), containers element-wise."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        return float('%.*g' % (digits, v)) if np.isfinite(v) else None
    if isinstance(v, (np.floating, np.integer)):
        return _sig(v.item(), digits)
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    return str(v)


def compact_line(full, detail_name='bench_detail.json'):
    """The one stdout line: the contract's keys, `config` as shapes + one-word modes, `roofline` as scalars, `cpu_baseline`,
    `reference_cpu`, `secondary` as [config, mode, ms_per_step, frac] rows, and for N > 1 the scalars a scaling figure needs to explain
    itself (multi_gpu, scaling_base).  Everything else is in the detail file.  Raises if the line would exceed COMPACT_LIMIT."""
    c = {k: full[k] for k in _TOP_KEYS if k in full}
    c['config'] = {k: full['config'][k] for k in _CONFIG_KEYS if k in full['config']}
    c['roofline'] = {k: full['roofline'][k] for k in _ROOFLINE_KEYS if k in full['roofline']}
    if 'cpu_baseline' in full:
        c['cpu_baseline'] = {k: full['cpu_baseline'][k] for k in ('value', 'unit', 'cores', 'kind', 'sample')}
    if 'reference_cpu' in full:
        c['reference_cpu'] = {k: full['reference_cpu'][k] for k in ('value', 'unit', 'cores', 'source')}
    if 'secondary' in full:
        c['secondary_fields'] = list(SECONDARY_FIELDS)
        c['secondary'] = [[_sig(e[k], 5) for k in SECONDARY_FIELDS] for e in full['secondary']]
    if full.get('child_errors'):
        c['child_errors'] = sorted(full['child_errors'])
    m = full.get('multi_gpu')
    if m is not None:
        cm = {'per_rank_ms_per_step': _sig(m['per_rank_ms_per_step'], 5),
              'host_blocked_ms_per_step_max': max(m['gather_wait_ms']['host_blocked_per_step']),
              'stream_stalled_ms_per_step_max': max(m['gather_wait_ms']['stream_stalled_per_step']),
              'bytes_received_by_rank0_per_step': m['bytes_received_by_rank0_per_step'], 'rank0_receive_GBps': m['rank0_receive_GBps'],
              'collectives_timed': m.get('collectives_timed'), 'warmup_steps_actually_run': m.get('warmup_steps_actually_run')}
        if 'span_tuning' in m:
            st = m['span_tuning']
            cm['span_tuning'] = {'chosen': st['chosen'], 'admissible': st['admissible'], 'stall_limit': st['stall_limit'],
                                 'fields': ['span_steps', 'ms_per_step', 'stall_frac'],
                                 'candidates': [[r['span_steps'], _sig(r['ms_per_step'], 5), _sig(r['stall_frac'], 3)] for r in st['candidates']]}
        if 'ideal_vs_n1_headline' in m:
            cm['ideal_vs_n1_headline'] = m['ideal_vs_n1_headline']
            cm['n1_headline_mode_ms_per_step'] = m['n1_headline_mode']['ms_per_step']
        if 'learner_rebuild' in m:
            cm['learner_rebuild_ms_per_step'] = m['learner_rebuild']['ms_per_step']
        c['multi_gpu'] = cm
    if 'scaling_base' in full:
        c['scaling_base'] = {k: full['scaling_base'][k] for k in ('value_per_gpu', 'unit', 'ms_per_step', 'efficiency')}
    c['detail'] = detail_name
    line = json.dumps(_sig(c), separators=(',', ':'), allow_nan=False)
    if len(line.encode()) > COMPACT_LIMIT:
        raise ValueError('bench.py: the result line is %d bytes, more than COMPACT_LIMIT = %d' % (len(line.encode()), COMPACT_LIMIT))
    return line


def write_detail(full, path):
    """The full record: to `path` (best effort: the tree may be read-only) and as one line to stderr."""
    text = json.dumps(full, default=lambda o: o.item() if isinstance(o, (np.floating, np.integer)) else str(o))
    try:
        with open(path, 'w') as f:
            f.write(text + '\n')
    except OSError as exc:
        print('bench.py: could not write %s (%s); the detail follows on stderr only' % (path, exc), file=sys.stderr)
    print(DETAIL_PREFIX + text, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--config', default='cfg3', choices=sorted(CONFIGS))
    ap.add_argument('--n-envs', type=int, default=0, help='envs per GPU (default: the config\'s)')
    ap.add_argument('--launch', default='auto', choices=['auto', 'span', 'step', 'graph'],
                    help='how the K steps are enqueued.  span: fmarl_step_span -- runs of steps as ONE launch in which '
                         'every workgroup walks its own envs through time, the episode-ending step a launch of its own (actions '
                         'come from a tape: BASELINE\'s random-action rollout); step: one fmarl_step call per step (what a policy in '
                         'the loop gets); graph: one hipGraph replay per episode (N=1).  auto = span for every N (the same launch '
                         'mode whatever the number of GPUs), except for nav_fairassign_fairrew_formation_graph (episodes end inside its '
                         'span; the per-episode record a learner rank needs is packed between launches: step)')
    ap.add_argument('--span-steps', type=int, default=0, help='most steps per span launch (0 = auto: up to the episode end for N = 1; '
                    '%d when the records of a run are gathered to a learner rank -- a run\'s records can only leave when its launch has '
                    'ended, and the last run\'s gather of the timed region is exposed)' % GATHER_SPAN_STEPS)
    ap.add_argument('--slots', default='ring', choices=['ring', 'same'],
                    help='ring: every step writes its own time slot of an episode-long ring of (T, n, ...) arrays (the trajectory exists '
                         'afterwards; every byte is written once per pass); same: every step overwrites ONE output set (round 3\'s mode)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gather', action='store_true', help='multi-GPU: skip the RCCL trajectory gather')
    ap.add_argument('--no-scaling-base', action='store_true', help='N > 1: skip the untimed-exchange pass that measures scaling_base')
    ap.add_argument('--record-path', action='store_true', help='N=1: write the step records / episode records as the '
                    'multi-GPU run does (the gather itself is a no-op with one rank)')
    ap.add_argument('--sync-reset', action='store_true', help='do not stage the next episode on a side stream')
    ap.add_argument('--tune-placement', action='store_true', help='--slots same only: keep the fastest (node_obs, adj) allocation pair of a few '
                    '(RolloutEngine tune_placement) instead of the first allocations')
    ap.add_argument('--graph', action='store_true', help='same as --launch graph')
    ap.add_argument('--eager', action='store_true', help='same as --launch step')
    ap.add_argument('--pipeline', type=int, default=1, help='step the env batch as this many sub-batches on their own streams '
                    '(fair_marl_amd.PipelinedRollout: the tail of one sub-batch\'s step kernel overlaps the head of the next one\'s); '
                    'one fmarl_step call per sub-batch and step, one output set per sub-batch, no trajectory gather in this mode')
    ap.add_argument('--rccl-selftest', action='store_true', help='N=1: open an RCCL process group of ONE rank and run the step / '
                    'episode gathers through it inside the timed loop (the nccl code path on a one-GPU box); implies --record-path')
    ap.add_argument('--no-secondary', action='store_true', help='N=1: skip the other BASELINE configs after the headline region')
    ap.add_argument('--detail', default=os.path.join(ROOT, 'bench_detail.json'), help='where the full record goes (prose, tables, per-rank rows, '
                    'the full secondary entries); stdout carries the compact line only')
    ap.add_argument('--secondary-child', default=None, choices=sorted(CONFIGS), help=argparse.SUPPRESS)   # (internal: run_secondary_children)
    ap.add_argument('--no-span-tuning', action='store_true', help='N > 1: runs of GATHER_SPAN_STEPS steps instead of choosing the run length '
                    'during the warm-up (SPAN_TUNE_CANDIDATES)')
    ap.add_argument('--learner-rebuild', type=int, default=0, metavar='K', help='rank 0 rebuilds node_obs / adj of K ranks\' gathered '
                    'steps (fmarl_rebuild_graph) inside the timed loop -- what a learner has to do with what arrives, since node_obs / '
                    'adj never travel; reported separately in the line (needs the gather: N > 1, --record-path or --rccl-selftest)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='N > 1: "nccl" is RCCL over xGMI (one GPU per rank); "gloo" rehearses the same exchange with '
                         'ranks sharing GPUs (local rank modulo the device count) -- its rate is not a result')
    args = ap.parse_args()
    if args.secondary_child:
        return secondary_child(args.secondary_child)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started by hand: launch one rank per GPU as a child job (nothing here has touched the GPU yet)
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        raise SystemExit(subprocess.call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                                          '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
                                          '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)'
                         % (args.gpus, world))
    spec = CONFIGS[args.config]
    cpu = None
    if world == 1 and not args.no_cpu_baseline:   # first: nothing in this process has initialised the GPU yet
        workers = max(1, min(16, os.cpu_count() or 1))
        if 'rocprof' in os.environ.get('LD_PRELOAD', '').lower() or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ):
            workers = 1   # under rocprofv3 the preloaded tool has already initialised the GPU: no exec from here
        try:
            cpu = cpu_baseline(spec['env'], spec['cpu_envs'], spec['cpu_episodes'], workers)
        except Exception as exc:   # e.g. no process spawning on this host: fall back to one in-process worker
            print('bench.py: multi-process cpu baseline failed (%s); using one process' % exc, file=sys.stderr)
            cpu = cpu_baseline(spec['env'], spec['cpu_envs'], spec['cpu_episodes'], 1)
    # the other configs' secondary entries: child processes, before this one touches the GPU (not under rocprofv3, whose preloaded
    # tool has initialised the GPU already: no process may be started from here then -- the entries run in-process, and say so)
    want_secondary = (world == 1 and not args.no_secondary and args.pipeline <= 1 and not args.graph and not args.eager and args.launch == 'auto'
                      and args.slots == 'ring' and args.span_steps == 0 and not (args.record_path or args.n_envs or args.rccl_selftest))
    under_profiler = 'rocprof' in os.environ.get('LD_PRELOAD', '').lower() or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)
    child_entries, child_errors, child_wall = {}, {}, 0.0
    if want_secondary and not under_profiler:
        child_entries, child_errors, child_wall = run_secondary_children(args.config)
    # stdout carries exactly one JSON line.  Libraries write there too -- the GPU boxes export NCCL_DEBUG=VERSION and RCCL
    # prints a five-line version banner to stdout when the first communicator is created -- so from here on file descriptor 1
    # is stderr, and the line goes to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    device = torch.device('cuda', local_rank if args.backend == 'nccl' else local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group('gloo')
    elif args.rccl_selftest:
        import socket
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=device)
        args.record_path = True

    cfg = fm.EnvConfig(**spec['env'])
    n_envs = args.n_envs or spec['n_envs']
    K, W = args.steps, args.warmup
    ep = cfg.episode_length
    workload = spec['workload'] % n_envs
    fnav_sc = cfg.scenario_name == 'nav_fairassign_fairrew_formation_graph'
    gather = (world > 1 or args.record_path) and not args.no_gather
    launch, span_steps = launch_plan('graph' if args.graph else ('step' if args.eager else args.launch), args.pipeline, cfg.scenario_name,
                                     gather, args.span_steps, ep)
    tune_span = world > 1 and gather and launch == 'span' and args.span_steps == 0 and not args.no_span_tuning
    span_len = [span_steps]      # steps per span launch (the tuning pass changes it)
    plain = [False]              # True: spans as an N = 1 run enqueues them (time slots only, no record buffers, nothing exchanged)
    if fnav_sc and gather and launch == 'span':
        raise SystemExit('bench.py: nav_fairassign_fairrew_formation_graph ends episodes inside a span; with the trajectory gather its '
                         'per-episode records must be packed between launches: use --launch step (the default)')
    if launch == 'graph':
        if world > 1:
            raise SystemExit('bench.py: --launch graph is a single-GPU mode')
        K, W = max(ep, K // ep * ep), (W + ep - 1) // ep * ep   # whole episodes
    pipe = None
    slots = args.slots if (args.pipeline <= 1 and launch != 'graph') else 'same'   # (graph replays and sub-batch engines write one output set)
    if slots == 'ring' and not _use_ring(cfg, n_envs, 'ring', device):
        print('bench.py: an episode-long ring of time slots does not fit this GPU\'s free memory: every step overwrites one output set (--slots same)',
              file=sys.stderr)
        slots = 'same'
    epb_hint = ((SAME_SLOT_EPB if slots == 'same' else SPAN_EPB).get(args.config, 0)) if launch == 'span' and not args.n_envs else 0
    if args.pipeline > 1:
        if args.rccl_selftest:
            raise SystemExit('bench.py: --pipeline does not combine with --rccl-selftest')
        if world > 1 and not args.no_gather:   # never a multi-GPU number that silently dropped the exchange
            raise SystemExit('bench.py: --pipeline has no trajectory gather; for N > 1 say so with --no-gather')
        gather = False
        pipe = fm.PipelinedRollout(cfg, n_envs, k=args.pipeline, device=device, seed=1, env_offset=rank * n_envs,
                                   async_reset=not args.sync_reset, tune_placement=0)
        eng = pipe.engines[0]
    else:
        eng = fm.RolloutEngine(cfg, n_envs, device=device, seed=1, env_offset=rank * n_envs, async_reset=not args.sync_reset,
                               tune_placement=(6 if args.tune_placement and slots == 'same' else 0), emit_graph_record=gather,
                               envs_per_workgroup=epb_hint)
    ring = fm.OutputRing(eng, ep) if slots == 'ring' else None   # step t writes slot t mod episode_length
    depth = 3   # record buffers in rotation: a gather has two runs' (steps') time before its buffer is written again
    # the learner rebuilds node_obs / adj from obs + a record gathered once per episode (navigation_graph) plus, for the two
    # formation scenarios, a per-step record of the step's scenario state (RolloutEngine.step_record_words)
    episodes = True
    graph_words = eng.step_record_words if eng.emit_graph_record else 0
    tg = sg = None
    if gather and launch == 'span':   # the records of a whole run of steps travel in one collective (SURVEY section 8 e)
        sg = tg = SpanGather(min(ep, max((span_steps,) + (SPAN_TUNE_CANDIDATES if tune_span else ()))), n_envs, cfg.N, cfg.obs_dim, device, dst=0, depth=depth, force_collective=args.rccl_selftest,
                             episode_words=eng.episode_record_words if episodes else 0, graph_words=graph_words, timing=True)
    elif gather:
        tg = TrajectoryGather(n_envs, cfg.N, cfg.obs_dim, device, dst=0, depth=depth, force_collective=args.rccl_selftest,
                              episode_words=eng.episode_record_words if episodes else 0, graph_words=graph_words, timing=True)
    # --learner-rebuild K: the learner rank turns K ranks' gathered steps back into node_obs / adj inside the timed loop
    rebuild_ranks, rebuild_events, lr_node, lr_adj = [], [], None, None
    if args.learner_rebuild:
        if not gather:
            raise SystemExit('bench.py: --learner-rebuild needs the trajectory gather (N > 1, --record-path or --rccl-selftest)')
        if rank == 0:
            peers = list(range(1, world)) + [0]          # the peers first; this rank's own record if K asks for more
            rebuild_ranks = peers[:max(1, min(args.learner_rebuild, world))]
            lr_node = torch.empty(n_envs, cfg.N, cfg.E, cfg.node_feat, dtype=torch.float32, device=device)
            lr_adj = torch.empty(n_envs, cfg.E, cfg.E, dtype=torch.float32, device=device)
    # Output sets.  The compact record of a step (obs, reward, done, + the graph record) goes into the exchange's buffers when
    # there is one; node_obs / adj / info go to the step's time slot (ring) or to the engine's one set of them (same).
    set_cache = {}

    def record_set(key, slot, obs, rew, done, graph):
        if (key, slot) not in set_cache:
            kw = {}
            if ring is not None:
                kw = dict(node_obs=ring.node_obs[slot], adj_env=ring.adj_env[slot], info_planes=ring.info_planes[slot] if ring.info_planes is not None else None)
            set_cache[(key, slot)] = eng.new_output_set(obs=obs, reward=rew, done=done, graph_record=graph if eng.emit_graph_record else None, **kw)
        return set_cache[(key, slot)]

    # synthetic action tape: int32 U{0..4} per (step, env, agent), resident in HBM before timing; one episode long (step t
    # reads entry t mod episode_length: a span's steps are consecutive entries)
    g = torch.Generator(device=device)
    g.manual_seed(1000 + rank)
    tape_len = ep
    tape = torch.randint(0, 5, (tape_len, n_envs, cfg.N), device=device, generator=g, dtype=torch.int32)

    inject_error = bool(os.environ.get('FMARL_BENCH_INJECT_GATHER_ERROR'))   # test hook: tests/test_hip_parity.py
    chunk = [0]          # runs of steps enqueued so far (span mode)
    chunk_ended = {}     # run index -> its last step ended an episode
    exchange = [True]    # False during the scaling_base pass: same launches, same record writes, nothing submitted
    submits = [0]        # records handed to the exchange so far

    def rebuild(obs_r, epi_r, graph_r):
        eng.rebuild_graph(obs_r, epi_r, node_obs=lr_node, adj_env=lr_adj, step_record=graph_r if eng.emit_graph_record else None)

    def run_steps(first, count):   # one fmarl_step call per step
        if pipe is not None:
            for t in range(first, first + count):
                pipe.step(tape[t % tape_len], auto_reset=True)
            return
        for t in range(first, first + count):
            if gather:
                r = tg.record(t)
                eng.use_outputs(record_set(t % depth, t % ep if ring is not None else 0, r.obs, r.reward, r.done, r.graph))
            elif ring is not None:
                eng.use_outputs(ring.sets[t % ep])
            eng.step(tape[t % tape_len], auto_reset=True)
            if gather and exchange[0]:
                if inject_error and submits[0] == 1 and rank == world - 1:
                    raise RuntimeError('injected gather error (FMARL_BENCH_INJECT_GATHER_ERROR)')
                tg.submit(t)
                submits[0] += 1
                if rebuild_ranks and t > first:
                    # step t - 1 has arrived (or is awaited here) while step t is in flight; its episode record is the latest
                    # one submitted -- this step's, if it started an episode, goes out below
                    got, graphs, epi = tg.gathered(t - 1), tg.gathered_graph(t - 1), tg.gathered_episode()
                    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ea.record()
                    for r in rebuild_ranks:
                        rebuild(got[r][0], epi[r], graphs[r])
                    eb.record()
                    rebuild_events.append((ea, eb, 1))
                if episodes and eng.episode_started:   # same steps on every rank (lockstep episodes)
                    eng.pack_episode(out=tg.episode_record())
                    tg.submit_episode()
        if gather:
            tg.finish()

    def run_spans(first, count, taper=True, finish=True):   # fmarl_step_span: runs of steps (span_schedule)
        use_gather = gather and not plain[0]
        t = first
        for k in span_schedule(first, count, ep, span_len[0], taper and use_gather):
            off = t % ep
            c = chunk[0]
            strides = None
            if use_gather:
                rec = sg.span_record(c)          # waits for the gather that last used this buffer (run c - depth)
                obs0, rew0, done0, graph0 = rec.first_step_buffers()
                eng.use_outputs(record_set(c % depth, off if ring is not None else 0, obs0, rew0, done0, graph0))
                strides = dict(rec.strides)
                if ring is not None:
                    strides.update(node_obs=ring.strides['node_obs'], adj=ring.strides['adj'], info=ring.strides['info'])
            elif ring is not None:
                eng.use_outputs(ring.sets[off])
                strides = ring.strides
            eng.step_span(tape[off:off + k], strides=strides)
            if use_gather and exchange[0]:
                if inject_error and submits[0] == 1 and rank == world - 1:
                    raise RuntimeError('injected gather error (FMARL_BENCH_INJECT_GATHER_ERROR)')
                sg.submit_span(c, k)
                submits[0] += 1
                chunk_ended[c] = bool(eng.episode_started)
                if rebuild_ranks and c - 1 in chunk_ended:
                    # run c - 1 has arrived (or is awaited here) while run c is in flight.  Its steps belong to the episode whose
                    # record was current during that run -- except its last step when that ended an episode: the observation
                    # behind an auto-reset belongs to the new episode, whose record went out right after run c - 1
                    got = sg.gathered_span(c - 1)
                    epi_new, epi_old = sg.gathered_episode(), sg.gathered_episode(back=1 if chunk_ended[c - 1] else 0)
                    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ea.record()
                    steps = got[0][0].shape[0]
                    for j in range(steps):
                        epi = epi_new if (j == steps - 1 and chunk_ended[c - 1]) else epi_old
                        for r in rebuild_ranks:
                            rebuild(got[r][0][j], epi[r], got[r][3][j] if got[r][3] is not None else None)
                    eb.record()
                    rebuild_events.append((ea, eb, steps))
                if episodes and eng.episode_started:   # the run ended an episode: the new one's record, once
                    eng.pack_episode(out=sg.episode_record())
                    sg.submit_episode()
            chunk[0] = c + 1
            t += k
        if use_gather and finish:
            sg.finish()

    run = run_spans if launch == 'span' else run_steps
    if pipe is not None:
        pipe.reset()
    else:
        eng.reset()
    # the box's write ceiling, first reading: before the warm-up, on the buffer the run writes (second reading after the timed region)
    ceil_dst = ring.node_obs if (pipe is None and ring is not None) else None   # (read where the run writes time slots: secondary_line)
    ceil0 = (store_ceiling(device, algorithmic_bytes(cfg) * n_envs * cfg.N, min(span_steps, ep - 1) if launch == 'span' else 1, dst=ceil_dst)
             if rank == 0 and ceil_dst is not None else None)
    if gather and episodes and pipe is None:   # the first episode's record
        eng.pack_episode(out=tg.episode_record())
        tg.submit_episode()
    counts_eager = None
    if launch == 'graph':
        # the per-kernel hipEvents cannot live inside a captured graph: the dominant kernel is timed in an eager pass
        # over whole episodes first, then the same steps run as graph replays inside the timed region
        eng.profile_enable(2 * tape_len)
        c_before_eager = eng.launch_counts()
        run_steps(0, 2 * tape_len)
        torch.cuda.synchronize(device)
        kernel_ms_eager = eng.profile_read(with_steps=True)
        eng.profile_enable(0)
        counts_eager = (c_before_eager, eng.launch_counts())
        episode_graph = eng.capture_steps(tape, lockstep=True)   # whole episodes from phase 0: one reset per episode in the graph

        def run(first, count):   # noqa: F811 -- whole episodes, one launch each
            for _ in range(count // tape_len):
                episode_graph.replay()
    try:
        run(0, W)
        torch.cuda.synchronize(device)
    except RuntimeError as exc:
        # A trajectory exchange that cannot run is a FAILED multi-GPU measurement, never a number without the exchange.
        print('bench.py: rank %d: rollout / trajectory gather failed: %s' % (rank, exc), file=sys.stderr, flush=True)
        sys.exit(1)
    # N > 1: the base the scaling is measured against -- the SAME K steps in the same launch mode with the same record writes, only
    # nothing is handed to the exchange: what every GPU does when it is alone.  Then untimed steps up to the same episode phase.
    first = W
    span_tuning = None
    if tune_span:
        # The run length, chosen here: for every candidate SPAN_TUNE_RUNS runs with the exchange on, timed
        # like the region itself (barrier, max over ranks); a run's wait for the gather of the run `depth` before it is the stall
        # the exchange causes in steady state (the closing waits of a pass are not counted: the timed region tapers its last runs).
        table = []
        for L in [c for c in SPAN_TUNE_CANDIDATES if c <= ep]:
            span_len[0] = L
            n_tune = SPAN_TUNE_RUNS * L
            sg.host_wait_s, sg.waits = 0.0, 0
            torch.cuda.synchronize(device)
            sg.stream_wait_ms()
            dist.barrier()
            tt = time.perf_counter()
            run_spans(first, n_tune, taper=False, finish=False)
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - tt
            row = torch.tensor([dt / n_tune * 1e3, sg.stream_wait_ms() / n_tune, sg.host_wait_s / n_tune * 1e3], dtype=torch.float64, device=device)
            sg.finish()
            torch.cuda.synchronize(device)
            dist.all_reduce(row, op=dist.ReduceOp.MAX)
            ms, stream_ms, host_ms = (float(v) for v in row.cpu())
            table.append({'span_steps': L, 'ms_per_step': ms, 'stream_stalled_per_step': stream_ms, 'host_blocked_per_step': host_ms,
                          'stall_frac': max(stream_ms, host_ms) / ms})
            first += n_tune
        pick = pick_span_length(table)
        span_len[0] = span_steps = pick['span_steps']
        span_tuning = {'candidates': table, 'chosen': span_steps, 'runs_per_candidate': SPAN_TUNE_RUNS, 'rule': pick_span_length.__doc__,
                       # how many candidates passed the stall rule (0: the rule's fallback branch chose -- the lowest time per step of all)
                       'admissible': sum(1 for r in table if r['stall_frac'] < SPAN_TUNE_STALL), 'stall_limit': SPAN_TUNE_STALL,
                       'steps_run': sum(SPAN_TUNE_RUNS * r['span_steps'] for r in table)}
        pad = (W - first) % ep      # back to the episode phase the region would have started at without the tuning pass
        run_spans(first, pad, taper=False)
        first += pad
        if episodes:
            torch.cuda.synchronize(device)
    base_elapsed = headline_elapsed = None
    if world > 1 and gather and launch != 'graph' and not args.no_scaling_base:
        exchange[0] = False
        dist.barrier()
        torch.cuda.synchronize(device)
        tb = time.perf_counter()
        run(first, K)
        torch.cuda.synchronize(device)
        dist.barrier()
        tb1 = torch.tensor([time.perf_counter() - tb], dtype=torch.float64, device=device)
        dist.all_reduce(tb1, op=dist.ReduceOp.MAX)
        base_elapsed = float(tb1.item())
        first += K
        pad = (-K) % ep
        run(first, pad)
        first += pad
        if launch == 'span':
            # and the N = 1 HEADLINE's launch mode on the same GPUs: whole-episode spans into the time slots, no record buffers --
            # what the driver's N = 1 line measures and divides the N = 8 value by.  ideal_vs_n1_headline = N x (the base above) /
            # (this): the scaling a perfect exchange would show against that line (runs of five steps pay a head and a tail per run)
            plain[0], keep_len = True, span_len[0]
            span_len[0] = ep
            dist.barrier()
            torch.cuda.synchronize(device)
            tb = time.perf_counter()
            run(first, K)
            torch.cuda.synchronize(device)
            dist.barrier()
            tb1 = torch.tensor([time.perf_counter() - tb], dtype=torch.float64, device=device)
            dist.all_reduce(tb1, op=dist.ReduceOp.MAX)
            headline_elapsed = float(tb1.item())
            first += K
            run(first, pad)
            first += pad
            plain[0], span_len[0] = False, keep_len
        exchange[0] = True
        if episodes:   # the episode that is under way now: its record (the pass above submitted none)
            eng.pack_episode(out=tg.episode_record())
            tg.submit_episode()
        torch.cuda.synchronize(device)
    if launch != 'graph':
        for e in (pipe.engines if pipe is not None else [eng]):
            e.profile_enable(K)
    if tg is not None:   # the warm-up's waits do not count
        tg.host_wait_s, tg.waits = 0.0, 0
        tg.stream_wait_ms()
    rebuild_events.clear()
    counts0 = eng.launch_counts()
    chunk_start = chunk[0]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    try:
        run(first, K)
    except RuntimeError as exc:
        print('bench.py: rank %d: rollout / trajectory gather failed: %s' % (rank, exc), file=sys.stderr, flush=True)
        sys.exit(1)
    torch.cuda.synchronize(device)
    own_elapsed = time.perf_counter() - t0   # this rank's own K steps, before it waits for the others
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    counts1 = eng.launch_counts()
    ranks_seen = 1
    per_rank = None
    if dist.is_initialized():   # one row per rank: [own ms per step, host ms blocked in gather waits, stream ms stalled in them]
        row = torch.tensor([own_elapsed / K * 1e3, (tg.host_wait_s if tg else 0.0) / K * 1e3,
                            (tg.stream_wait_ms() if tg else 0.0) / K], dtype=torch.float64, device=device)
        rows = [torch.zeros_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
        per_rank = torch.stack(rows).cpu().numpy()
    elif tg is not None:
        per_rank = np.array([[own_elapsed / K * 1e3, tg.host_wait_s / K * 1e3, tg.stream_wait_ms() / K]])
    if dist.is_initialized():
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        ones = torch.ones(1, dtype=torch.int32, device=device)   # every rank adds itself: the collective really spans the job
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())
        if ranks_seen != world:
            print('bench.py: all_reduce saw %d ranks, expected %d' % (ranks_seen, world), file=sys.stderr, flush=True)
            sys.exit(1)
        if gather and rank == 0:   # the last gathered record really holds every rank's rows (rank r's envs start at r * n_envs)
            if sg is not None:
                got = sg.gathered_span(chunk[0] - 1)
                bad = len(got) != world or any(tuple(o.shape[1:]) != (n_envs, cfg.N, cfg.obs_dim) for o, _, _, _ in got)
            else:
                got = tg.gathered(first + K - 1)
                bad = len(got) != world or any(tuple(o.shape) != (n_envs, cfg.N, cfg.obs_dim) for o, _, _ in got)
            if bad:
                print('bench.py: gathered record has the wrong shape', file=sys.stderr, flush=True)
                sys.exit(1)
    if launch == 'graph':
        kernel_ms, kernel_steps = kernel_ms_eager
    else:
        kernel_ms, kernel_steps = [], []
        for e in (pipe.engines if pipe is not None else [eng]):
            ms, st = e.profile_read(with_steps=True)
            kernel_ms += ms
            kernel_steps += st

    if rank == 0:
        agents = n_envs * cfg.N
        # dominant kernel = the step kernel.  A launch on an episode-end step whose reset observation is written by separate
        # reset launches emits nothing itself (state + reward bytes only); with the staged reset of navigation_graph the launch
        # that ends an episode also commits the next one and emits its first observation (step_end_kernel), so every launch
        # writes the full outputs.  Which of the two happened is read from the library's launch counters, not assumed.
        resets = sum(1 for t in range(first, first + K) if (t + 1) % ep == 0)   # (fairnav episodes may also end earlier, env by env)
        ca, cb = counts_eager if launch == 'graph' else (counts0, counts1)
        sub = max(1, args.pipeline)
        if pipe is not None:
            ca = (0, 0, 0, 0)   # sub-batch engines: their totals since creation (includes the warm-up's launches: same mix)
            cb = tuple(sum(e.launch_counts()[i] for e in pipe.engines) for i in range(4))
        bytes_per_step = launch_bytes(cfg, agents // sub, ca, cb) * sub      # mean algorithmic bytes of one step of all envs
        folded = cb[1] - ca[1]
        kernel_ms, kernel_steps = np.asarray(kernel_ms, dtype=np.float64), np.asarray(kernel_steps, dtype=np.float64)
        span_launches = kernel_steps > 1
        bytes_moved = None
        if launch == 'span' and span_launches.any():
            # the dominant kernel is the span kernel: a launch covers a run of steps, all of which emit
            k_avg_ms = float(kernel_ms[span_launches].mean())
            steps_per_launch = float(kernel_steps[span_launches].mean())
            bytes_per_launch = agents * algorithmic_bytes(cfg) * steps_per_launch
            bytes_moved = moved_bytes(cfg, agents, steps_per_launch)
            kernel_name = {'fair_graph_formation': 'formation_span_kernel',
                           'nav_fairassign_fairrew_formation_graph': 'fairnav_span_kernel'}.get(cfg.scenario_name, 'step_span_small_kernel' if _small_batch(cfg, eng) else 'step_span_kernel')
        else:
            k_avg_ms = float(kernel_ms.mean()) if kernel_ms.size else float('nan')
            steps_per_launch = 1.0
            bytes_per_launch = bytes_per_step / sub   # a launch steps one sub-batch
            kernel_name = KERNEL_NAMES.get(cfg.scenario_name, ('step_small_kernel' if _small_batch(cfg, eng) else 'step_kernel') + (' / step_end_kernel' if folded else ''))
        achieved = bytes_per_launch / (k_avg_ms * 1e-3) / 1e9
        traffic = None
        tkey = '%s/%s%s' % (args.config, launch, '-ring' if slots == 'ring' else '')
        tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tpath) and pipe is None:   # (the committed counters are per STEP over all envs of the profiled run)
            with open(tpath) as f:
                entry = json.load(f).get(tkey, {})
            if entry.get('n_envs') == n_envs and entry.get('hbm_bytes_per_step'):
                traffic = entry['hbm_bytes_per_step'] * steps_per_launch
        step_k_ms = float(kernel_ms.sum() / max(1.0, kernel_steps.sum()))   # all step launches of the region / its steps
        emis_ms = emission_only_ms(eng) if pipe is None else None
        launch_text = {
            'span': 'fmarl_step_span: runs of steps (up to the episode end%s) as ONE launch each in which every workgroup walks its own envs through '
                    'time (%d envs per workgroup; %s), the step that ends an episode %s: %d launches for the %d timed steps'
                    % ('' if span_steps >= ep else ', at most %d steps' % span_steps, eng.envs_per_workgroup,
                       'state in registers, static entities in LDS between the steps',
                       'inside the span (this scenario resets its ended envs in the step)' if fnav_sc else 'as a launch of its own', len(kernel_ms), K),
            'step': ('%d sub-batches of %d envs on their own streams, one fmarl_step call per sub-batch and step'
                     % (args.pipeline, n_envs // args.pipeline) if pipe is not None else 'one fmarl_step call per step'),
            'graph': 'one hipGraph replay per episode of %d steps (kernel_avg_ms from an eager pass before the timed region)' % ep}[launch]
        if gather and sg is not None:
            exchange_text = 'one gather per run of steps (<= %d steps: %d B per agent-step, back to back)' % (min(ep, span_steps), StepRecord.bytes_per_agent_step(cfg.obs_dim, graph_words))
        else:
            exchange_text = 'gather of obs/reward/done to rank 0 every step, %d B per agent-step' % StepRecord.bytes_per_agent_step(cfg.obs_dim, graph_words)
        ring_bytes = ring.nbytes if ring is not None else 0
        # the box's write ceiling, on the pages the kernel itself wrote (the ring's node_obs slots, else the engine's node_obs);
        # then the engine and the time slots go (the secondary lines build their own)
        ceil = None
        if pipe is None:
            if ceil_dst is not None:
                ceil = store_ceiling(device, algorithmic_bytes(cfg) * agents, steps_per_launch, dst=ceil_dst, earlier=ceil0)
            eng.close()
            set_cache.clear()
            arena = ring   # (its arrays serve the headline config's secondary entries)
            del eng, ring, tape, ceil_dst
            lr_node = lr_adj = None
            torch.cuda.empty_cache()
        out = {
            'metric': 'env agent-steps/sec (n_envs x n_agents / wall-s), %s random-action rollout' % cfg.scenario_name,
            'value': world * agents * K / elapsed, 'unit': 'agent-steps/s', 'n_gpus': world, 'n_ranks_seen': ranks_seen,
            'steps': K, 'warmup': W,
            'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': workload, 'n_envs_per_gpu': n_envs, 'n_agents': cfg.N,
                       'n_entities': cfg.E, 'episode_length': ep, 'auto_resets_timed': resets,
                       'arithmetic': 'f64 state, contact forces and statistics; f32 outputs (obs, node_obs, adj, reward, info)',
                       'launch_mode': launch, 'launch': launch_text, 'span_steps': (min(ep, span_steps) if launch == 'span' else None),
                       # one word each in the compact line; the sentences behind them under *_text in the detail
                       'slots': slots,
                       'slots_text': ('ring: every step writes its own time slot of an episode-long ring of (%d, n, ...) arrays (%.1f GB): the trajectory '
                                      'exists afterwards, every byte is written once per pass' % (ep, ring_bytes / 1e9) if slots == 'ring'
                                      else 'same: every step overwrites one output set'),
                       'reset': 'synchronous' if args.sync_reset or cfg.scenario_name != 'navigation_graph' else 'staged',
                       'reset_text': ('synchronous' if args.sync_reset or cfg.scenario_name != 'navigation_graph'
                                      else 'next episode staged on a side stream, committed and observed by the launch that ends the episode'
                                           ' (%d of %d episode ends folded)' % (folded, cb[1] - ca[1] + cb[2] - ca[2])),
                       'exchange': ('none' if not gather else
                                    (('rccl' + ('-selftest' if world == 1 else '') if args.backend == 'nccl' and dist.is_initialized()
                                      else ('records-only' if world == 1 else 'gloo')) + '-gather-per-' + ('run' if sg is not None else 'step'))),
                       'exchange_text': (('RCCL' + (' (process group of one rank: self-test)' if world == 1 else '') if args.backend == 'nccl' and dist.is_initialized()
                                     else ('record writes only, no process group' if world == 1 else 'gloo (rehearsal)')) + ' ' + exchange_text
                                    + ((' + %d B per env with EVERY step (goals, landmarks, obstacles, walls: this scenario\'s episodes end env by env, '
                                        'so the record is re-packed and gathered whenever an env may have been reset)' if fnav_sc else
                                        ' + %d B per env once per episode (goals, landmarks, obstacles, walls)')
                                       % (4 * episode_words_of(cfg)) if episodes else '')) if gather else 'none'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'traffic_source': ('profiles/pmc_traffic.json [%s]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier '
                                            'run of this command, replayed here, not measured in this run' % tkey) if traffic is not None else None,
                         'kernel': kernel_name,
                         'kernel_avg_ms': k_avg_ms, 'kernel_launches': int(span_launches.sum()) if steps_per_launch > 1 else int(kernel_ms.size),
                         'kernel_steps_per_launch': steps_per_launch,
                         'step_kernels_ms_per_step': step_k_ms,
                         'slots': slots,
                         # every step in a slot of its own (each byte written once per launch) / one output set rewritten every step:
                         # the headline's own figure under its name, the other mode's from the 300-step secondary entry (below)
                         'frac_distinct_slots': achieved / HBM_PEAK_GBS if slots == 'ring' else None,
                         'frac_same_slot': achieved / HBM_PEAK_GBS if slots == 'same' else None,
                         # the formula's bytes with the state and the static entity words counted once per span launch (they stay on the chip)
                         'bytes_moved_per_launch': bytes_moved,
                         'frac_bytes_moved': (bytes_moved / (k_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if bytes_moved else None,
                         # the box's write ceiling: the best pure 16-byte store stream over the byte count of the dominant kernel's
                         # launch, per step; frac_of_box_ceiling = that / the step kernels' time per step (<= 1 by construction)
                         'store_ceiling_ms': ceil['ms_per_step'] if ceil else None,
                         'store_ceiling': ceil,
                         'frac_of_box_ceiling': (ceil['ms_per_step'] / step_k_ms if ceil else None),
                         'emission_only_ms': emis_ms,   # one fmarl_rebuild_graph launch on the last step's buffers (rounds 2-3's "ceiling")
                         'algorithmic_bytes_per_launch': bytes_per_launch,
                         'algorithmic_bytes_per_agent_step': algorithmic_bytes(cfg)},
        }
        if pipe is not None:
            # launches of different sub-batches overlap: a launch's own duration (above) understates what the chip delivers;
            # the whole-job figure is the algorithmic bytes of one step of ALL envs over the timed region's time per step
            job = bytes_per_step / (elapsed / K) / 1e9
            out['roofline']['overlap'] = {'sub_batches': args.pipeline, 'job_achieved': job, 'job_frac': job / HBM_PEAK_GBS,
                                          'basis': 'algorithmic bytes per step of all envs / timed region per step; '
                                                   'kernel_avg_ms is the duration of ONE sub-batch launch while others run'}
        if per_rank is not None:
            # one run must explain itself: every rank's own time per step, what it waited for the exchange, what the learner
            # rank receives and what it costs to turn that back into node_obs / adj
            rec_bytes = (sg.spans[0].stride if sg is not None else tg.records[0].flat.numel()) if tg is not None else 0
            recv_bytes = rec_bytes * (world - 1)
            out['multi_gpu'] = {
                'per_rank_ms_per_step': [float(v) for v in per_rank[:, 0]],
                'gather_wait_ms': {'host_blocked_per_step': [float(v) for v in per_rank[:, 1]],
                                   'stream_stalled_per_step': [float(v) for v in per_rank[:, 2]],
                                   'note': 'time inside the gather\'s waits (for the buffer of the run / step two back, at the end of the '
                                           'region): an RCCL wait stalls the compute stream, a gloo wait blocks the host'},
                'bytes_gathered_per_step': rec_bytes * world, 'bytes_received_by_rank0_per_step': recv_bytes,
                'rank0_receive_GBps': recv_bytes / (elapsed / K) / 1e9,
                'collectives': ('one per run of steps (%d in the timed region)' % (chunk[0] - chunk_start) if sg is not None else 'one per step'),
                'collectives_timed': (chunk[0] - chunk_start) if sg is not None else K,
                # every untimed step before the timed region: --warmup, the run-length tuning passes, the scaling_base passes and their
                # pads back to the episode phase (with --warmup 5 and the tuner on: 5 + 112 + ...), so `warmup` is not mistaken for it
                'warmup_steps_actually_run': first,
                'episode_record_bytes_per_rank': 4 * episode_words_of(cfg) * n_envs if episodes else 0,
                'episode_record_gathers_per_step': (1.0 if fnav_sc and sg is None else 1.0 / ep) if episodes else 0.0}
            if rebuild_ranks:
                torch.cuda.synchronize(device)
                ms = [a.elapsed_time(b) for a, b, _ in rebuild_events]
                nst = sum(k for _, _, k in rebuild_events)
                out['multi_gpu']['learner_rebuild'] = {
                    'ranks_rebuilt_per_step': rebuild_ranks, 'steps_rebuilt': nst,
                    'ms_per_step': float(np.sum(ms) / nst) if nst else None,
                    'agent_steps_rebuilt_per_s': (len(rebuild_ranks) * agents * nst / (np.sum(ms) * 1e-3)) if nst else None,
                    'note': 'rank 0 rebuilds node_obs / adj of these ranks\' gathered steps (fmarl_rebuild_graph) inside the timed loop, '
                            'on the same stream as its own step kernels; the reference ships node_obs / adj instead '
                            '(onpolicy/envs/env_wrappers.py:983-996)'}
        if span_tuning is not None:
            out['multi_gpu']['span_tuning'] = span_tuning
        if base_elapsed is not None and headline_elapsed is not None:
            out['multi_gpu']['ideal_vs_n1_headline'] = world * headline_elapsed / base_elapsed
            out['multi_gpu']['n1_headline_mode'] = {
                'value_per_gpu': agents * K / headline_elapsed, 'ms_per_step': headline_elapsed / K * 1e3,
                'note': 'the same %d steps as whole-episode spans into the time slots with no record buffers and nothing exchanged (the launch mode of '
                        'the N = 1 line), max over the %d ranks; ideal_vs_n1_headline = n_gpus x scaling_base.value_per_gpu / this: what a perfect '
                        'exchange would show against the N = 1 line -- the rest of a shortfall is the exchange (scaling_base.efficiency)' % (K, world)}
        if base_elapsed is not None:
            # weak scaling against what the same GPUs do when nothing is exchanged: the same K steps, launch mode, span length and
            # record writes on every rank, timed the same way (barrier, max over ranks) just before the timed region
            base = agents * K / base_elapsed
            out['scaling_base'] = {'value_per_gpu': base, 'unit': 'agent-steps/s', 'ms_per_step': base_elapsed / K * 1e3,
                                   'efficiency': out['value'] / (world * base),
                                   'basis': 'the same %d steps in the same launch mode (%s%s, slots %s) with the same record writes and no exchange, '
                                            'max over the %d ranks; efficiency = value / (n_gpus x value_per_gpu).  The N = 1 line of this '
                                            'bench carries runs of %d steps as `secondary` entry (%s, span%d) next to its own headline'
                                            % (K, launch, ' of at most %d steps' % span_steps if launch == 'span' else '', slots, world,
                                               GATHER_SPAN_STEPS, args.config, GATHER_SPAN_STEPS)}
        if cpu is not None:
            out['cpu_baseline'] = cpu
        if args.config in REFERENCE_CPU:
            out['reference_cpu'] = REFERENCE_CPU[args.config]
        if want_secondary and pipe is None:
            t_sec = time.perf_counter()
            # the headline config again over 300 steps (as spans, one launch per step, in runs of GATHER_SPAN_STEPS, into one output
            # set) on the headline's own arrays; every other BASELINE config that fits one GPU: measured in child processes before
            # this one touched the GPU (run_secondary_children), here only where that was not possible
            out['secondary'] = []
            for name, mode in SECONDARY:
                if name in child_entries:
                    if mode == [m for n, m in SECONDARY if n == name][0]:
                        out['secondary'] += child_entries[name]
                    continue
                if name != args.config and arena is not None:   # the headline config's entries come first: then its slots go
                    arena = None
                    torch.cuda.empty_cache()
                e = secondary_line(name, mode, device, arena=arena if name == args.config else None)
                e['process'] = ('the headline\'s process, on the headline\'s arrays' if name == args.config else
                                'the headline\'s process, after its ring was freed (%s): late entries vary with the device\'s allocation history'
                                % ('under a profiler no child process may be started' if under_profiler else child_errors.get(name, 'child failed')))
                out['secondary'].append(e)
            out['secondary_wall_s'] = time.perf_counter() - t_sec + child_wall
            if child_errors:   # configs whose child process failed or timed out (their entries then ran in this process: ADVICE round 5)
                out['child_errors'] = child_errors
            for e in out['secondary']:   # the other slot mode of the headline config, over 300 steps
                if e['config'] == args.config and e['mode'] == 'span-same':
                    out['roofline']['frac_same_slot'] = e['frac']
                    out['roofline']['same_slot_ms_per_step'] = e['kernel_avg_ms']
                if e['config'] == args.config and e['mode'] == 'span':
                    out['roofline']['frac_distinct_slots_300_steps'] = e['frac']
        sys.stdout.flush()
        line = compact_line(out, os.path.basename(args.detail))
        write_detail(out, args.detail)
        os.write(result_fd, (line + '\n').encode())
    if dist.is_initialized():
        dist.destroy_process_group()


def episode_words_of(cfg):
    """32-bit words per env of the episode record (fmarl_episode_record_words: goals + landmarks + obstacles + 6 per wall)."""
    return 2 * cfg.N + 2 * (cfg.num_landmarks + cfg.num_obstacles) + 6 * cfg.num_walls


if __name__ == '__main__':
    main()
