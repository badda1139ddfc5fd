"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol declared in
include/fmarl.h, state-layout queries (host code only), spaces / specs / lazy infos, sharding and the
trajectory gather over gloo with world_size 2.  No kernel is launched."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


@pytest.fixture(scope='module')
def lib():
    graft.build()
    from fair_marl_amd import _lib
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, 'include', 'fmarl.h')).read()
    declared = set(re.findall(r'\b(fmarl_[a-z_]+)\s*\(', hdr))
    from fair_marl_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert getattr(raw, name) is not None


def test_state_layout_and_config_validation(lib):
    import fair_marl_amd as fm
    from fair_marl_amd import _lib
    cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
    c = cfg.to_c(65536, seed=3, env_offset=65536)
    total = lib.fmarl_state_bytes(C.byref(c))
    prev_end = 0
    for fid, name in enumerate(_lib.FIELD_NAMES):
        off, cnt, dt = C.c_size_t(), C.c_size_t(), C.c_int()
        assert lib.fmarl_state_field(C.byref(c), fid, C.byref(off), C.byref(cnt), C.byref(dt)) == 0
        assert off.value % 256 == 0 and off.value >= prev_end
        prev_end = off.value + cnt.value * _lib.DTYPE_BYTES[dt.value]
    assert prev_end <= total
    off, cnt, dt = C.c_size_t(), C.c_size_t(), C.c_int()
    lib.fmarl_state_field(C.byref(c), _lib.F_AGENT_POS, C.byref(off), C.byref(cnt), C.byref(dt))
    assert cnt.value == 65536 * 32 * 2 and dt.value == _lib.DTYPE_F64
    assert lib.fmarl_state_field(C.byref(c), 99, None, None, None) != 0
    # invalid configs are refused with a message, not a crash
    bad = fm.EnvConfig(num_agents=3, num_landmarks=4).to_c(8)
    h = C.c_void_p()
    assert lib.fmarl_create(C.byref(bad), C.byref(h)) != 0
    assert b'num_landmarks' in lib.fmarl_last_error()
    assert lib.fmarl_state_bytes(C.byref(bad)) == 0
    ok = C.c_void_p()
    assert lib.fmarl_create(C.byref(c), C.byref(ok)) == 0 and ok.value
    assert lib.fmarl_destroy(ok) == 0


def test_entry_points_validate_arguments_on_the_host(lib):
    """Bad arguments are refused with FMARL_EINVAL + a message before anything is launched."""
    import fair_marl_amd as fm
    c = fm.EnvConfig(num_agents=3, num_landmarks=3).to_c(8)
    h = C.c_void_p()
    assert lib.fmarl_create(C.byref(c), C.byref(h)) == 0
    assert lib.fmarl_step(h, None, None, None, None, 1, None) == 1 and b'null' in lib.fmarl_last_error()
    assert lib.fmarl_reset(h, None, None, None, None) == 1
    assert lib.fmarl_init_state(h, None, None) == 1
    assert lib.fmarl_profile_enable(h, -1) == 1
    assert lib.fmarl_lexifair(None, None, 4, 3, None) == 1
    buf = (C.c_double * 8)()
    assert lib.fmarl_lexifair(buf, buf, 1, 65, None) == 1 and b'64' in lib.fmarl_last_error()
    assert lib.fmarl_cost_matrix(None, None, None, 1, 1, 1, None) == 1
    assert lib.fmarl_update_graph(None, None, None, None, 1, 1, 1.0, None) == 1
    assert lib.fmarl_edge_count(None, None, 1, 1, 1.0, 1, None) == 1
    assert lib.fmarl_info_means(None, None, 1, 1, 2.5, None) == 1
    assert lib.fmarl_state_changed(None) == 1
    base, cookie = C.c_void_p(), C.c_void_p()
    assert lib.fmarl_ring_alloc(0, 4, 0, C.byref(base), C.byref(cookie)) == 1 and lib.fmarl_ring_alloc(1 << 20, 0, 0, C.byref(base), C.byref(cookie)) == 1
    assert lib.fmarl_ring_free(None) == 1 and lib.fmarl_ring_stats(None) == 1
    stats = (C.c_uint64 * 8)()
    assert lib.fmarl_ring_stats(stats) == 0 and stats[0] >= stats[1] and stats[2] >= stats[3] and stats[4] == 8 << 40
    assert lib.fmarl_store_stream(None, 4096, 0, 0, 1, 0, None) == 1 and lib.fmarl_store_stream(buf, 64, 1, 100, 1, 0, None) == 1
    assert lib.fmarl_store_stream(buf, 1 << 20, 2, 4096, 2, 0, None) == 1 and b'coprime' in lib.fmarl_last_error()   # 256 chunks, order 2: not a permutation
    assert lib.fmarl_pack_episode(h, None, None, None) == 1 and lib.fmarl_rebuild_graph(h, None, None, 4, None, None, None) == 1
    assert lib.fmarl_episode_started(None) == 0 and lib.fmarl_episode_started(h) == 0
    assert lib.fmarl_episode_record_words(C.byref(c)) == 2 * 3 + 2 * (3 + 3) + 0
    assert lib.fmarl_get_state(h, None, 0, None, None) == 1 and lib.fmarl_set_state(h, None, 99, None, None) == 1
    assert lib.fmarl_destroy(h) == 0
    for kw, msg in ((dict(num_agents=0, num_landmarks=0), b'num_agents'), (dict(num_agents=3, num_landmarks=3, num_walls=3), b'num_walls'),
                    (dict(num_agents=3, num_landmarks=3, episode_length=0), b'episode_length')):
        bad = fm.EnvConfig(**kw).to_c(4)
        assert lib.fmarl_create(C.byref(bad), C.byref(h)) == 1 and msg in lib.fmarl_last_error()
    glob = fm.EnvConfig(num_agents=3, num_landmarks=3, graph_feat_type='global').to_c(4)
    assert glob.flags & 2
    glob.num_walls = 1   # the Python side refuses this earlier; the C side must too
    assert lib.fmarl_create(C.byref(glob), C.byref(h)) == 1 and b'global' in lib.fmarl_last_error()
    one = fm.EnvConfig(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=1, num_landmarks=1).to_c(4)
    assert lib.fmarl_create(C.byref(one), C.byref(h)) == 1   # the scenario needs a second-nearest goal
    form = fm.EnvConfig(scenario_name='fair_graph_formation', num_agents=10, num_landmarks=1).to_c(16)
    assert form.num_walls == 2 and lib.fmarl_state_bytes(C.byref(form)) > 0


def test_engine_refuses_to_run_without_gpu():
    import fair_marl_amd as fm
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        fm.RolloutEngine(fm.EnvConfig(num_agents=3, num_landmarks=3), 8)


def test_specs_spaces_and_unsupported_scenarios():
    import argparse
    import fair_marl_amd as fm
    args = argparse.Namespace(scenario_name='navigation_graph', num_agents=32, num_landmarks=32, num_obstacles=8,
                              num_walls=0, episode_length=25)
    spec = fm.GraphMPEEnv(args)
    spec.seed(1 + 3 * 1000)
    assert spec.seed_value == 3001 and spec.n == 32
    assert spec.observation_space[0].shape == (7,) and spec.observation_space[0].__class__.__name__ == 'Box'
    assert spec.share_observation_space[0].shape == (224,)
    assert spec.node_observation_space[0].shape == (72, 11) and spec.adj_observation_space[0].shape == (72, 72)
    assert spec.action_space[0].__class__.__name__ == 'Discrete' and spec.action_space[0].n == 5
    args.scenario_name = 'simple_spread_graph'
    with pytest.raises(NotImplementedError):
        fm.GraphMPEEnv(args)
    args.scenario_name = 'navigation_graph'
    args.graph_feat_type = 'global'     # navigation_graph.py:1058-1077: 7 absolute columns
    assert fm.GraphMPEEnv(args).node_observation_space[0].shape == (72, 7)
    args.num_walls = 1                   # the reference's global features know no walls (ValueError there too)
    with pytest.raises(ValueError):
        fm.GraphMPEEnv(args)
    args.num_walls, args.scenario_name = 0, 'nav_fairassign_fairrew_formation_graph'
    with pytest.raises(NotImplementedError):
        fm.GraphMPEEnv(args)


def test_lazy_infos_match_reference_structure():
    from fair_marl_amd.infos import INFO_KEYS, LazyInfos
    rec = np.arange(2 * 3 * 14, dtype=np.float64).reshape(2, 3, 14)
    infos = LazyInfos(rec)
    assert len(infos) == 2 and len(infos[1]) == 3
    d = infos[1][2]
    assert list(d)[0] == 'individual_reward' and set(d) == set(INFO_KEYS)
    assert d['Dist_to_goal'] == rec[1, 2, 0] and d['individual_reward'] == rec[1, 2, 13]
    assert [len(e) for e in infos] == [3, 3]
    assert [a['Time_taken'] for a in infos[0]] == list(rec[0, :, 8])
    f = LazyInfos(rec, 'fair_graph_formation')[0][1]
    assert len(f) == 12 and f['Formation_dist'] == rec[0, 1, 9] and 'Time_mean' not in f


def test_agent_count_limits_are_named_at_the_boundary(lib):
    """The reference's Python loops take any number of agents; the device kernels do not.  The limits stand next to
    FmarlConfig.num_agents in include/fmarl.h and in INTEGRATION.md, and a config beyond them is refused with a text that
    names the supported range (never a crash or a silently wrong launch)."""
    import fair_marl_amd as fm
    hdr = open(os.path.join(ROOT, 'include', 'fmarl.h')).read()
    integ = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    for text in ('navigation_graph 1..64', 'fair_graph_formation 1..32', 'nav_fairassign_fairrew_formation_graph 2..32'):
        assert text in hdr and text in integ, text
    h = C.c_void_p()
    for kw, want in ((dict(num_agents=65, num_landmarks=65), b'navigation_graph is built for num_agents in 1..64'),
                     (dict(scenario_name='fair_graph_formation', num_agents=33, num_landmarks=1), b'fair_graph_formation is built for num_agents in 1..32'),
                     (dict(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=33, num_landmarks=33),
                      b'nav_fairassign_fairrew_formation_graph is built for num_agents in 2..32'),
                     (dict(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=1, num_landmarks=1),
                      b'nav_fairassign_fairrew_formation_graph is built for num_agents in 2..32')):
        c = fm.EnvConfig(**kw).to_c(4)
        assert lib.fmarl_create(C.byref(c), C.byref(h)) != 0 and want in lib.fmarl_last_error(), lib.fmarl_last_error()
        assert lib.fmarl_state_bytes(C.byref(c)) == 0 and want in lib.fmarl_last_error()
    for kw in (dict(num_agents=64, num_landmarks=64), dict(scenario_name='fair_graph_formation', num_agents=32, num_landmarks=1),
               dict(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=32, num_landmarks=32)):
        assert lib.fmarl_state_bytes(C.byref(fm.EnvConfig(**kw).to_c(4))) > 0   # the limits themselves are supported


def test_shard_range_partitions_exactly():
    from fair_marl_amd.sharding import shard_range
    for n, w in ((524288, 8), (10, 3), (7, 8), (65536, 1)):
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _gather_worker(rank, world, port, q, n_total=12, N=3, D=7):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from fair_marl_amd.sharding import TrajectoryGather, shard_range
    T = 5
    lo, hi = shard_range(n_total, world, rank)
    tg = TrajectoryGather(hi - lo, N, D, 'cpu', dst=0, depth=2, episode_words=4, graph_words=9)
    ok = True
    for t in range(T):
        if t % 2 == 0:                           # a new 'episode' every second step: its record travels once
            tg.episode_record().copy_(torch.arange(lo, hi, dtype=torch.int32).view(-1, 1) * 10 + t // 2)
            tg.submit_episode()
        rec = tg.record(t)                       # waits for the gather that used this buffer (t - 2)
        env = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1)
        rec.obs.copy_((env + 0.25 * t).expand(hi - lo, N, D))   # (exact in float32 up to 2^19 envs)
        rec.reward.copy_((env[:, :, 0] + 0.5 * t).expand(hi - lo, N))
        rec.done.fill_(t % 2)
        rec.graph.copy_((torch.arange(lo, hi, dtype=torch.int32).view(-1, 1, 1) * 1000 + t).expand(hi - lo, N, 9))
        tg.submit(t)
        if t >= 1:                               # consume step t-1 on the learner rank while t is in flight
            if rank == 0:
                for r, (obs, rew, done) in enumerate(tg.gathered(t - 1)):
                    l2, h2 = shard_range(n_total, world, r)
                    e = torch.arange(l2, h2, dtype=torch.float32)
                    ok &= bool((obs[:, 0, D - 1] == e + 0.25 * (t - 1)).all()) and bool((rew[:, N - 1] == e + 0.5 * (t - 1)).all())
                    ok &= bool((done == (t - 1) % 2).all()) and obs.shape == (h2 - l2, N, D)
                for r, gr in enumerate(tg.gathered_graph(t - 1)):
                    l2, h2 = shard_range(n_total, world, r)
                    ok &= gr.shape == (h2 - l2, N, 9) and bool((gr[:, N - 1, 8] == torch.arange(l2, h2, dtype=torch.int32) * 1000 + t - 1).all())
                for r, ep in enumerate(tg.gathered_episode()):     # the record of the episode step t belongs to
                    l2, h2 = shard_range(n_total, world, r)
                    ok &= ep.shape == (h2 - l2, 4) and bool((ep[:, 3] == torch.arange(l2, h2, dtype=torch.int32) * 10 + t // 2).all())
    tg.finish()
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_trajectory_gather_world_size_2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def _span_gather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from fair_marl_amd.sharding import SpanGather, shard_range
    n_total, N, D, G, Tmax = 14, 3, 7, 5, 6
    lo, hi = shard_range(n_total, world, rank)
    sg = SpanGather(Tmax, hi - lo, N, D, 'cpu', dst=0, depth=2, episode_words=4, graph_words=G)
    ok = True
    runs = [4, 6, 1, 5, 6]                       # run lengths (<= Tmax), as spans that end at episode ends give them
    env = torch.arange(lo, hi, dtype=torch.float32).view(1, -1, 1)
    for c, k in enumerate(runs):
        rec = sg.span_record(c)                  # waits for the gather that used this buffer (run c - 2)
        assert rec.strides['obs'] * 4 == rec.stride and rec.strides['done'] == rec.stride and rec.stride % 16 == 0
        step = torch.arange(k, dtype=torch.float32).view(-1, 1, 1)
        rec.obs[:k].copy_((env * 64 + step + 0.25 * c).unsqueeze(-1).expand(k, hi - lo, N, D))
        rec.reward[:k].copy_((env + 100 * step).expand(k, hi - lo, N))
        rec.done[:k].copy_(((step + c) % 2).to(torch.uint8).expand(k, hi - lo, N))
        rec.graph[:k].copy_((env * 1000 + step + c).to(torch.int32).unsqueeze(-1).expand(k, hi - lo, N, G))
        o0, r0, d0, g0 = rec.first_step_buffers()   # what the engine's output set points at: step 0 of the run
        ok &= bool(torch.equal(o0, rec.obs[0])) and bool(torch.equal(g0, rec.graph[0])) and o0.is_contiguous() and d0.data_ptr() == rec.done[0].data_ptr()
        sg.submit_span(c, k)
        if c % 2 == 0:
            sg.episode_record().copy_(torch.arange(lo, hi, dtype=torch.int32).view(-1, 1) * 10 + c)
            sg.submit_episode()
        if c >= 1 and rank == 0:                 # consume run c - 1 on the learner rank while run c is in flight
            kk = runs[c - 1]
            for r, (obs, rew, done, graph) in enumerate(sg.gathered_span(c - 1)):
                l2, h2 = shard_range(n_total, world, r)
                e = torch.arange(l2, h2, dtype=torch.float32).view(1, -1)
                st = torch.arange(kk, dtype=torch.float32).view(-1, 1)
                ok &= obs.shape == (kk, h2 - l2, N, D) and bool((obs[:, :, N - 1, D - 1] == e * 64 + st + 0.25 * (c - 1)).all())
                ok &= bool((rew[:, :, 0] == e + 100 * st).all()) and bool((done[:, :, 1] == ((st + c - 1) % 2).to(torch.uint8)).all())
                ok &= graph.shape == (kk, h2 - l2, N, G) and bool((graph[:, :, 2, G - 1] == (e * 1000 + st + c - 1).to(torch.int32)).all())
            for r, ep in enumerate(sg.gathered_episode()):
                l2, h2 = shard_range(n_total, world, r)
                ok &= bool((ep[:, 3] == torch.arange(l2, h2, dtype=torch.int32) * 10 + (c // 2) * 2).all())
    sg.finish()
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_span_gather_world_size_2_gloo():
    """SpanGather (bench.py's default exchange for N > 1): the records of a whole run of steps back to back in one buffer --
    the strides a span launch writes through -- ONE gather per run, runs of different lengths, double-buffered, episode
    records beside them (two equal shards of 7 envs; gloo on the CPU)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_span_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_trajectory_gather_world_size_8_with_the_layout_of_config_5():
    """BASELINE config 5's partition: 524 288 envs over 8 ranks = shard_range(524288, 8, r), 65 536 contiguous envs each,
    every step's record and every episode's record gathered to rank 0 and checked there against the GLOBAL env index
    (gloo on the CPU; one agent with a one-float obs keeps eight ranks' buffers small -- the exchange code does not
    depend on the row width)."""
    from fair_marl_amd.sharding import shard_range
    assert [shard_range(524288, 8, r) for r in (0, 1, 7)] == [(0, 65536), (65536, 131072), (458752, 524288)]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gather_worker, args=(r, 8, port, q, 524288, 1, 1)) for r in range(8)]
    for p in procs:
        p.start()
    ok = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok


def test_bench_launch_plan_is_the_same_for_every_number_of_gpus():
    """bench.py decides how the K steps are enqueued without looking at the number of GPUs: spans for the
    scenarios that have a span kernel, steps for the third one; what a gather changes is the length of a run (its records leave
    when its launch has ended), never the mode."""
    import bench
    nav, fnav = 'navigation_graph', 'nav_fairassign_fairrew_formation_graph'
    alone, sharded = bench.launch_plan('auto', 1, nav, False, 0, 25), bench.launch_plan('auto', 1, nav, True, 0, 25)
    assert alone == ('span', 25) and sharded == ('span', bench.GATHER_SPAN_STEPS) and alone[0] == sharded[0]
    assert bench.launch_plan('auto', 1, 'fair_graph_formation', True, 0, 25)[0] == 'span'
    assert bench.launch_plan('auto', 1, fnav, False, 0, 25)[0] == bench.launch_plan('auto', 1, fnav, True, 0, 25)[0] == 'step'
    assert bench.launch_plan('step', 1, nav, True, 0, 25)[0] == 'step' and bench.launch_plan('span', 2, nav, False, 0, 25)[0] == 'step'
    assert bench.launch_plan('auto', 1, nav, True, 12, 25) == ('span', 12) and bench.launch_plan('auto', 1, nav, False, 99, 25) == ('span', 25)
    # the runs of a region: episode ends and the run length split it; with a gather the last stretch halves down to single steps
    assert bench.span_schedule(55, 30, 25, 5, True) == [5, 5, 5, 5, 5, 3, 1, 1] and bench.span_schedule(55, 30, 25, 5, False) == [5] * 6
    assert bench.span_schedule(5, 30, 25, 25, True) == [20, 5, 3, 1, 1] and bench.span_schedule(0, 50, 25, 25, False) == [25, 25]
    assert bench.span_schedule(5, 30, 25, 12, True) == [12, 8, 5, 3, 1, 1] and bench.span_schedule(7, 0, 25, 5, True) == []
    # the run length of an N > 1 run: the fastest admissible candidate (gather waits under 2 % of the time per step), else the fastest
    rows = [dict(span_steps=3, ms_per_step=1.40, stall_frac=0.0), dict(span_steps=5, ms_per_step=1.31, stall_frac=0.01),
            dict(span_steps=8, ms_per_step=1.28, stall_frac=0.05), dict(span_steps=12, ms_per_step=1.26, stall_frac=0.30)]
    assert bench.pick_span_length(rows)['span_steps'] == 5
    assert bench.pick_span_length([dict(r, stall_frac=0.5) for r in rows])['span_steps'] == 12
    assert bench.pick_span_length([dict(r, stall_frac=0.0) for r in rows])['span_steps'] == 12
    for L in bench.SPAN_TUNE_CANDIDATES:
        for first in (0, 5, 24):
            runs = bench.span_schedule(first, 30, 25, L, True)
            assert sum(runs) == 30 and max(runs) <= L and runs[-1] == 1
    # the formula of SURVEY section 8(d) and the state-once-per-launch variant of it
    import fair_marl_amd as fm
    cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
    assert bench.algorithmic_bytes(cfg) == 3966.0
    assert bench.moved_bytes(cfg, 1, 1) == pytest.approx(3966.0) and bench.moved_bytes(cfg, 1, 24) == pytest.approx(24 * 3966.0 - 23 * (96 + 10))


def worst_case_bench_record(world=8, n_secondary=None):
    """A full bench record as large as bench.py can make it: every `secondary` entry with its prose, the N = 8 `multi_gpu` block with the
    span-tuning table, the learner rebuild, scaling_base, child errors, long floats everywhere (tests/test_hip_parity.py checks the real one)."""
    import bench
    modes = list(bench.SECONDARY) if n_secondary is None else [('cfg3', 'span%d' % i) for i in range(n_secondary)]
    f = 1234567.890123456789
    prose = 'fmarl_step_span: one launch per run of steps between episode ends and at most 12 steps (8 envs per workgroup), the episode-ending step a launch of its own ' * 3
    sec = [dict(config=n, mode=m, workload=bench.CONFIGS.get(n, bench.CONFIGS['cfg3'])['workload'] % 65536, launch=prose, slots=prose, value=f * 1e3, unit='agent-steps/s',
                steps=300, warmup=650, ms_per_step=0.123456789012345, kernel='step_span_small_kernel + step_end_kernel', kernel_avg_ms=0.123456789012345,
                kernel_launches=300, envs_per_workgroup=8, frac=0.123456789012345, algorithmic_bytes_per_step=f * 1e4, store_ceiling_ms=1.15432029117237,
                store_ceiling_shape=prose, frac_of_box_ceiling=0.96045436629790, bound=prose, timing=prose, regime=prose, process=prose, frac_basis=prose)
           for n, m in modes]
    ranks = [1.2345678901234 + 0.001 * r for r in range(world)]
    table = [dict(span_steps=L, ms_per_step=1.2345678901234, stream_stalled_per_step=0.0123456789, host_blocked_per_step=0.00123456789, stall_frac=0.0123456789)
             for L in bench.SPAN_TUNE_CANDIDATES]
    return {
        'metric': 'env agent-steps/sec (n_envs x n_agents / wall-s), nav_fairassign_fairrew_formation_graph random-action rollout',
        'value': f * 1e4, 'unit': 'agent-steps/s', 'n_gpus': world, 'n_ranks_seen': world, 'steps': 20, 'warmup': 5, 'ms_per_step': 1.2046123456789,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': bench.CONFIGS['fnav10']['workload'] % 65536, 'n_envs_per_gpu': 65536, 'n_agents': 32, 'n_entities': 72, 'episode_length': 25,
                   'auto_resets_timed': 1, 'arithmetic': prose, 'launch_mode': 'span', 'launch': prose, 'span_steps': 12, 'slots': 'ring', 'slots_text': prose,
                   'reset': 'synchronous', 'reset_text': prose, 'exchange': 'rccl-selftest-gather-per-step', 'exchange_text': prose},
        'roofline': {'bound': 'hbm', 'achieved': 6975.759235113068, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.8719699043891335, 'traffic': 156068995734.2222,
                     'traffic_source': prose, 'kernel': 'step_span_small_kernel', 'kernel_avg_ms': 22.65399169921875, 'kernel_launches': 1,
                     'kernel_steps_per_launch': 19.0, 'step_kernels_ms_per_step': 1.2018481373786927, 'slots': 'ring', 'frac_distinct_slots': 0.8719699043891335,
                     'frac_same_slot': 0.7613204091838678, 'bytes_moved_per_launch': 154027425792.0, 'frac_bytes_moved': 0.8498912014991149,
                     'store_ceiling_ms': 1.1543202911723744, 'store_ceiling': {'streams': {prose: 1.0, prose + 'b': 2.0}, 'shape': prose},
                     'frac_of_box_ceiling': 0.9604543662979089, 'emission_only_ms': 1.2108381271362305, 'algorithmic_bytes_per_launch': 158028791808.0,
                     'algorithmic_bytes_per_agent_step': 3966.0, 'overlap': {'basis': prose}},
        'cpu_baseline': dict(value=511721.8709528825, unit='agent-steps/s', cores=16, kind='port',
                             sample='16 processes x 512 envs x 20 episodes incl. auto-resets (491520000 agent-steps), NumPy f64 oracle, slowest worker 29.6 s, 33.5 s wall incl. process start'),
        'reference_cpu': dict(bench.REFERENCE_CPU['cfg4']),
        'secondary': sec, 'secondary_wall_s': 25.123456789, 'child_errors': {n: prose for n in bench.SECONDARY_CHILD_ORDER},
        'multi_gpu': {'per_rank_ms_per_step': ranks, 'gather_wait_ms': {'host_blocked_per_step': ranks, 'stream_stalled_per_step': ranks, 'note': prose},
                      'bytes_gathered_per_step': 65536 * 32 * 33 * world, 'bytes_received_by_rank0_per_step': 65536 * 32 * 33 * (world - 1),
                      'rank0_receive_GBps': 370.123456789012, 'collectives': prose, 'collectives_timed': 8, 'warmup_steps_actually_run': 237,
                      'episode_record_bytes_per_rank': 4 * 144 * 65536, 'episode_record_gathers_per_step': 0.04,
                      'learner_rebuild': {'ranks_rebuilt_per_step': list(range(1, world)), 'steps_rebuilt': 29, 'ms_per_step': 9.87654321098, 'agent_steps_rebuilt_per_s': f, 'note': prose},
                      'span_tuning': {'candidates': table, 'chosen': 12, 'runs_per_candidate': 4, 'rule': prose, 'admissible': 0, 'stall_limit': 0.02, 'steps_run': 112},
                      'ideal_vs_n1_headline': 7.123456789012, 'n1_headline_mode': {'value_per_gpu': f * 1e3, 'ms_per_step': 1.2046123456789, 'note': prose}},
        'scaling_base': {'value_per_gpu': f * 1e3, 'unit': 'agent-steps/s', 'ms_per_step': 1.2345678901234, 'efficiency': 0.87654321098765, 'basis': prose},
    }


def test_bench_result_line_is_compact_and_strict_json(tmp_path, capsys):
    """The ONE stdout line of bench.py stays under bench.COMPACT_LIMIT bytes whatever the run (round 5's line grew to 21 KB and the driver's
    record came back `parsed: null`): built here from a worst-case record -- every secondary entry, the N = 8 multi_gpu block, the
    tuning table -- it parses strictly, carries the contract's keys, `roofline` and `cpu_baseline` as scalars and `secondary` as four-field
    rows; the prose and the tables are in the detail record, and a line that would not fit is an error, never a longer line."""
    import json
    import bench
    full = worst_case_bench_record()
    assert len(json.dumps(full)) > 20000            # (what used to be printed)
    line = bench.compact_line(full)
    assert len(line.encode()) <= bench.COMPACT_LIMIT == 6000 and '\n' not in line
    d = json.loads(line, parse_constant=lambda c: pytest.fail('non-strict JSON constant %s' % c))
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                'config', 'roofline', 'cpu_baseline'):
        assert key in d, key
    assert d['n_ranks_seen'] == 8 and d['value'] == pytest.approx(full['value'], rel=1e-5) and d['ms_per_step'] == pytest.approx(full['ms_per_step'], rel=1e-5)
    assert set(d['config']) == set(bench._CONFIG_KEYS) and all(' ' not in str(d['config'][k]) for k in ('launch_mode', 'slots', 'reset', 'exchange'))
    r = d['roofline']
    assert set(r) == set(bench._ROOFLINE_KEYS) and all(not isinstance(v, (dict, list)) for v in r.values())
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-5) and r['traffic'] == pytest.approx(full['roofline']['traffic'], rel=1e-5)
    assert set(d['cpu_baseline']) == {'value', 'unit', 'cores', 'kind', 'sample'} and d['reference_cpu']['value'] == 1716.0
    assert d['secondary_fields'] == ['config', 'mode', 'ms_per_step', 'frac'] and len(d['secondary']) == len(bench.SECONDARY) >= 18
    assert all(len(row) == 4 and row[:2] == list(nm) for row, nm in zip(d['secondary'], bench.SECONDARY))
    m = d['multi_gpu']
    assert len(m['per_rank_ms_per_step']) == 8 and m['warmup_steps_actually_run'] == 237
    assert m['span_tuning']['chosen'] == 12 and m['span_tuning']['admissible'] == 0 and len(m['span_tuning']['candidates']) == len(bench.SPAN_TUNE_CANDIDATES)
    assert d['scaling_base']['efficiency'] == pytest.approx(0.876543, rel=1e-5) and d['child_errors'] == sorted(bench.SECONDARY_CHILD_ORDER)
    # NaN / inf never reach the line (strict JSON has no spelling for them)
    full['roofline']['traffic'] = float('nan')
    assert json.loads(bench.compact_line(full))['roofline']['traffic'] is None
    # a record that cannot fit is refused
    with pytest.raises(ValueError):
        bench.compact_line(worst_case_bench_record(n_secondary=200))
    # the detail record: the file and one stderr line behind the prefix, both the full record
    path = str(tmp_path / 'detail.json')
    full = worst_case_bench_record()
    bench.write_detail(full, path)
    err = [l for l in capsys.readouterr().err.splitlines() if l.startswith(bench.DETAIL_PREFIX)]
    assert len(err) == 1 and json.loads(err[0][len(bench.DETAIL_PREFIX):]) == json.load(open(path)) == json.loads(json.dumps(full))


def test_hot_kernels_keep_their_register_and_scratch_budget():
    """Compiler remarks of the gfx950 build (tools/kres.sh, no GPU needed): the three step kernels must not spill the
    kernel-argument block to scratch memory (a by-reference use of Params that is not inlined costs 776 bytes per lane and
    a third of the throughput -- it happened once) and must keep the occupancy DESIGN.md section 4 counts on."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(['bash', os.path.join(root, 'tools', 'kres.sh')], capture_output=True, text=True, timeout=600).stdout
    rows = {}
    for line in out.splitlines():
        m = re.match(r'(\S+)\s+vgpr\s+(\d+) scratch\s+(\d+) occ (\d+)', line)
        if m:
            rows[m.group(1)] = tuple(int(x) for x in m.groups()[1:])
    def find(part):
        hits = [v for k, v in rows.items() if part in k]
        assert len(hits) == 1, (part, list(rows))
        return hits[0]
    # scratch = 0: besides its cost, a scratch load is a VMEM load on gfx9 -- its s_waitcnt vmcnt(0) also waits for every global
    # store issued before it (the generic node emission ran at 2/3 of its rate while 24 bytes of a row lived in scratch)
    for part, max_vgpr, max_scratch, min_occ in (('11step_kernelILi0', 96, 0, 5), ('16formation_kernelILb1ELi0', 120, 0, 4), ('16formation_kernelILb1ELi1', 120, 0, 4),
                                                 ('14fairnav_kernelILb1ELi256ELi0', 128, 0, 4),
                                                 # ... and with the agent / goal count a compile-time 3 (the shipped FA / FA+FR configuration): unrolled loops, fewer registers
                                                 ('14fairnav_kernelILb1ELi256ELi3', 104, 0, 4), ('19fairnav_span_kernelILi192ELi3', 152, 0, 3),
                                                 ('19fairnav_span_kernelILi256ELi3', 120, 0, 4),
                                                 ('17reset_emit_kernel', 96, 0, 5),
                                                 ('20rebuild_graph_kernel', 96, 0, 5), ('15step_end_kernelILi0', 112, 0, 4),
                                                 # the span kernels (the bench's default launch mode) and the learner-side gather: the latter once
                                                 # compiled to 179 VGPRs = two waves per SIMD for a copy kernel (profiles/archive/r3_notes.md)
                                                 ('16step_span_kernelILi0', 160, 0, 3), ('21formation_span_kernelILi0', 128, 0, 4),
                                                 # BASELINE config 4's shape as compile-time constants: 72 bytes of scratch and still 5 % faster than the generic one
                                                 # (0.218 -> 0.206 ms per step on one box, profiles/r6_shapes.md)
                                                 ('21formation_span_kernelILi1', 128, 80, 4),
                                                 ('11step_kernelILi2', 100, 0, 5), ('16step_span_kernelILi2', 168, 16, 3), ('17step_small_kernelILi1', 88, 0, 5),
                                                 ('22step_span_small_kernelILi1', 168, 0, 3),
                                                 ('23minibatch_gather_kernel', 112, 0, 4),
                                                 # three waves per workgroup (64 envs x 3 agents: the shipped configuration): 168 registers at four
                                                 # workgroups per CU; six doubles of the carried state are spilled around the emission and reloaded at
                                                 # the top of the next step (loop depth 1, none of it inside the emission loops: profiles/r5_notes.md)
                                                 ('19fairnav_span_kernelILi192ELi0', 168, 96, 3),   # (any OTHER shape of up to three agents: rarely run; the unrolled adj walk took it from 64 to 80 bytes)
                                                 # the four-wave form (fewer than 160 or more than 192 agent lanes per workgroup: small geometries; only up
                                                 # to three agents -- beyond, fmarl_step_span launches per step): no carry, the state through global memory
                                                 # between the steps at four waves per SIMD (round 5: the carry at 128 registers, 236 bytes of scratch;
                                                 # the carry at 168 registers = three waves per SIMD measured 20 % slower, profiles/r6_fnav_spans_by_n.txt)
                                                 ('19fairnav_span_kernelILi256ELi0', 128, 136, 4)):
        vgpr, scratch, occ = find(part)
        assert vgpr <= max_vgpr and scratch <= max_scratch and occ >= min_occ, (part, vgpr, scratch, occ)
