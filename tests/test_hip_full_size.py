"""-m gpu: oracle parity AT THE BENCHMARKED LAUNCH GEOMETRY, through episode ends.

Every fixture / small-batch test runs a handful of envs, for which the library picks one env per workgroup (or the test
forces a geometry on tiled fixtures, test_hip_parity.py GEOM).  Here the engines are the ones bench.py times: 65 536 envs,
the staged (asynchronous) reset, auto-reset on -- the full-batch envs-per-workgroup, the episode-ending launch that commits the
staged episode and re-emits (navigation_graph: step_end_kernel), the in-kernel resets of the two formation scenarios with
only SOME envs of a workgroup ending.  Two full episodes + 2 steps from nothing but the seed; at EVERY step obs / node_obs /
adj / reward / done / info of a strided sample of envs are compared with the oracle stepping exactly those envs from
PhiloxStream(seed, global env index, episode) (reference semantics: onpolicy/envs/env_wrappers.py:859-865,
multiagent/custom_scenarios/navigation_graph.py:212-262, nav_fairassign_fairrew_formation_graph.py:732-739), and the full
float64 state of the sample at the end.
"""
import numpy as np
import pytest
import torch

import fair_marl_amd as fm
from oracle import fairnav_oracle as fnv
from oracle import formation_oracle as fo
from oracle import nav_oracle as no
from oracle.philox import PhiloxStream

pytestmark = pytest.mark.gpu
OUT = dict(rtol=1e-5, atol=1e-5)     # north_star: outputs within 1e-5 in float32
STATE = dict(rtol=1e-9, atol=1e-9)   # float64 state after whole episodes (test_hip_parity.py)
DEV = 'cuda:0'
N_ENVS = 65536
FORM_INFO = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13]   # record slots of fo.INFO_KEYS

CASES = {
    # BASELINE config 2 at its own size (4 096 envs x 3 agents): the one config whose launch geometry nothing else shares -- the
    # library spreads the small batch over 512 workgroups of 8 envs (24 of 256 lanes carry agents), E * F = 99 leaves through the
    # generic row windows
    'cfg2': dict(mod=no, seed=36, stride=61, n=4096, kw=dict(num_agents=3, num_landmarks=3, num_obstacles=3)),
    # BASELINE config 3 (bench.py default): 8 envs per workgroup, wave-scan statistics, 16-byte row streaming, step_end_kernel
    'cfg3': dict(mod=no, seed=31, stride=1021, kw=dict(num_agents=32, num_landmarks=32, num_obstacles=8)),
    # the reference's own 10-agent scale (bench.py --config n10): odd row widths through the per-wave LDS windows
    'n10': dict(mod=no, seed=32, stride=1021, kw=dict(num_agents=10, num_landmarks=10, num_obstacles=3)),
    # walls (the end-cap force takes cos / sin from past / size instead of arcsin -- state, not output): agents
    # crowded between two walls for two episodes, every step against the oracle's asin / cos / sin, float64 state at 1e-9 at the end
    'navw': dict(mod=no, seed=35, stride=509, kw=dict(num_agents=6, num_landmarks=6, num_obstacles=2, num_walls=2)),
    # BASELINE config 4 (bench.py --config cfg4): 24 envs per workgroup, six per wave, in-kernel reset
    'cfg4': dict(mod=fo, seed=33, stride=1021,
                 kw=dict(scenario_name='fair_graph_formation', num_agents=10, num_landmarks=1, num_obstacles=3)),
    # bench.py --config fnav with a threshold at which agents reach goals, get `status` and envs end early at different steps
    # inside one workgroup (119 early ends in the sample; on this seed no sampled env reaches the state in which the reference
    # itself raises, nf:888-903)
    'fnav': dict(mod=fnv, seed=34, stride=331,
                 kw=dict(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=3, num_landmarks=3, num_obstacles=3,
                         goal_rew=30.0, collision_rew=30.0, min_dist_thresh=0.5)),
    # the same scenario at two agents: more than 64 envs per workgroup -- the placement teams' second ballot word (fairnav_place_teams) --
    # and 16-lane teams that hold 6 entities
    'fnav2': dict(mod=fnv, seed=37, stride=499,
                  kw=dict(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=2, num_landmarks=2, num_obstacles=2,
                          goal_rew=30.0, collision_rew=30.0, min_dist_thresh=0.5)),
    # ... at ten agents (BASELINE.md section 2's N = 10 row, bench.py --config fnav10): the per-step kernel's 10-agent shape instance,
    # 32-lane teams that hold 23 entities, fmarl_step_span launching per step
    'fnav10': dict(mod=fnv, seed=38, stride=1511,
                   kw=dict(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=10, num_landmarks=10, num_obstacles=3,
                           goal_rew=30.0, collision_rew=30.0)),   # (the default threshold: at larger ones ten agents reach the state in which the reference itself raises, nf:888-903)
    # ... with walls: the teams draw the wall orientations, test the padded wall boxes and store the wall records; the re-seated lanes
    # re-read the wall tables from the state
    'fnavw': dict(mod=fnv, seed=39, stride=997, n=16384,
                  kw=dict(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=4, num_landmarks=4, num_obstacles=2, num_walls=2,
                          goal_rew=30.0, collision_rew=30.0, min_dist_thresh=0.45)),
}


def make_oracle(mod, ocfg, sample, seed):
    streams = lambda e, ep: PhiloxStream(seed, int(sample[e]), ep)  # noqa: E731  (the GLOBAL env index keys the stream)
    cls = {no: no.OracleGraphVecEnv, fo: fo.OracleFormationVecEnv, fnv: fnv.OracleFairNavVecEnv}[mod]
    return cls(ocfg, len(sample), mode='subproc', streams=streams)


def check(mod, got, ref, msg):
    obs, ids, node, adj, rew, done, info = got
    np.testing.assert_allclose(obs, ref[0], err_msg=msg + ' obs', **OUT)
    np.testing.assert_allclose(node, ref[2], err_msg=msg + ' node_obs', **OUT)
    np.testing.assert_allclose(adj, ref[3][:, 0], err_msg=msg + ' adj', **OUT)
    np.testing.assert_allclose(rew, ref[4], err_msg=msg + ' reward', **OUT)
    assert np.array_equal(done.astype(bool), ref[5]), msg + ' done'
    np.testing.assert_allclose(info[..., FORM_INFO] if mod is fo else info, ref[6], err_msg=msg + ' info', **OUT)


def _geometries():
    """(case, envs per workgroup): the library's choice for every case + the geometry bench.py forces for spans that rewrite one
    output set (bench.SAME_SLOT_EPB, read from bench.py so the two cannot drift) + whatever bench.SPAN_EPB forces for spans into
    time slots."""
    import bench
    geo = [(case, 0) for case in CASES]
    for table in (bench.SAME_SLOT_EPB, bench.SPAN_EPB):
        for name, epb in table.items():
            if name in CASES and (name, epb) not in geo:
                geo.append((name, epb))
    return geo


@pytest.mark.parametrize('case,epb_hint', _geometries())
def test_full_batch_rollout_through_episode_ends_vs_oracle(case, epb_hint):
    c = CASES[case]
    mod, seed = c['mod'], c['seed']
    cfg = fm.EnvConfig(**c['kw'])
    ocfg = mod.Config(**{k: getattr(cfg, k) for k in mod.Config.__dataclass_fields__})
    n, N = c.get('n', N_ENVS), cfg.N
    sample = np.unique(np.concatenate([np.arange(0, n, c['stride']), [n - 1]]))
    eng = fm.RolloutEngine(cfg, n, device=DEV, seed=seed, async_reset=True, tune_placement=0, envs_per_workgroup=epb_hint)
    epb = eng.envs_per_workgroup
    assert epb > 1 and (epb_hint == 0 or epb == epb_hint) and len({int(e) % epb for e in sample}) == min(epb, len(sample)), 'sample must cover every position of a workgroup'
    orc = make_oracle(mod, ocfg, sample, seed)
    s = torch.as_tensor(sample, device=DEV)
    pick = lambda res: tuple(x[s].cpu().numpy() for x in res)  # noqa: E731
    obs, ids, node, adj = eng.reset()
    o = orc.reset()
    np.testing.assert_allclose(obs[s].cpu().numpy(), o[0], err_msg='reset obs', **OUT)
    np.testing.assert_allclose(node[s].cpu().numpy(), o[2], err_msg='reset node_obs', **OUT)
    np.testing.assert_allclose(eng.adj_env[s].cpu().numpy(), o[3][:, 0], err_msg='reset adj', **OUT)
    rs = np.random.RandomState(seed)
    T = cfg.episode_length
    ends = early = 0
    for t in range(2 * T + 2):
        a = rs.randint(0, 5, size=(n, N)).astype(np.int32)
        res = eng.step(torch.as_tensor(a, device=DEV), auto_reset=True)
        ref = orc.step(a[sample])
        got = pick((res[0], res[1], res[2], eng.adj_env, res[4], res[5], res[6]))
        check(mod, got, ref, '%s step %d' % (case, t))
        d = ref[5].all(axis=1)
        ends += int(d.sum())
        early += int((d & ((t + 1) % T != 0)).sum())
    assert ends >= 2 * len(sample)
    if case.startswith('fnav'):
        if case != 'fnav10':   # (ten agents never all stop early at the default threshold: their envs end with the episode, all at once)
            assert early > (50 if case == 'fnav' else 5), early   # envs of one workgroup really ended at different steps
        if case == 'fnav2':
            assert epb > 64, epb   # the second ballot word of the placement teams
    else:
        assert eng.phase == 2      # lockstep kept: the episode ends went through the folded / in-kernel path
    got = eng.get_state()
    for k in mod.State.FIELDS:
        if k != 'time':
            np.testing.assert_allclose(got[k][sample], getattr(orc.st, k), err_msg='%s end state %s' % (case, k), **STATE)
    # size-independent properties of the whole batch after the last step
    adj_env = eng.adj_env
    assert torch.equal(adj_env, adj_env.transpose(1, 2)) and (torch.diagonal(adj_env, dim1=1, dim2=2) == 0).all()
    assert torch.isfinite(eng.node_obs).all() and torch.isfinite(eng.obs).all() and torch.isfinite(eng.reward).all()
    if mod is not fo:
        gm = eng.field('goal_match')
        assert torch.equal(torch.sort(gm, dim=1).values, torch.arange(N, device=DEV, dtype=torch.int32).expand(n, N))
    ep = eng.field('episode')
    assert int(ep.min()) >= 3   # the hidden make_world reset + reset() + two episode ends


@pytest.mark.parametrize('case,epb_hint', _geometries())
def test_full_batch_step_span_equals_step_by_step(case, epb_hint):
    """fmarl_step_span at the benchmarked size: 65 536 envs, two episodes + 3 steps from one tape, against an engine stepping
    the same tape one launch per step (the path the test above pins against the oracle): final state and outputs bit for bit.
    fnav: all 53 steps are ONE launch (fairnav_span_kernel), with envs ending their episodes early at different steps inside it."""
    c = CASES[case]
    cfg = fm.EnvConfig(**c['kw'])
    n, N, T = c.get('n', N_ENVS), cfg.N, 2 * cfg.episode_length + 3
    a = fm.RolloutEngine(cfg, n, device=DEV, seed=c['seed'], tune_placement=0)
    b = fm.RolloutEngine(cfg, n, device=DEV, seed=c['seed'], tune_placement=0, envs_per_workgroup=epb_hint)
    assert epb_hint == 0 or b.envs_per_workgroup == epb_hint
    gen = torch.Generator(device=DEV); gen.manual_seed(9)
    tape = torch.randint(0, 5, (T, n, N), device=DEV, generator=gen, dtype=torch.int32)
    a.reset(); b.reset()
    for t in range(T):
        a.step(tape[t])
    b.rollout(tape)                      # mode 'span' by default
    torch.cuda.synchronize()
    assert b.launch_counts()[0] == T and a.phase == b.phase == (-1 if case.startswith('fnav') else 3)
    if case == 'fnav':
        assert int((a.field('episode') > a.field('episode').min()).sum()) > 100, 'envs must have ended episodes early, at different steps'
    for k in ('obs', 'node_obs', 'adj_env', 'reward', 'done', 'info'):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    for k in a._fields:
        if not k.startswith(('stage_', 'internal_')) and k != 'reset_flag':
            assert torch.equal(a.field(k), b.field(k)), k


@pytest.mark.parametrize('case', ['cfg3', 'cfg4', 'n10', 'fnav', 'cfg2'])
def test_full_batch_spans_into_time_slots_vs_oracle(case):
    """bench.py's headline mode at its own size and geometry: 65 536 envs, every step of an episode written to its own time slot
    of an OutputRing by ONE span launch + the episode-ending launch (step_span with per-step strides), then a second pass over
    the ring.  Every slot of both passes -- obs / node_obs / adj / reward / done / info of a strided sample covering every
    position of a workgroup -- against the oracle stepping those envs (reference: every step's outputs are delivered,
    onpolicy/envs/env_wrappers.py:988-996), and the slots of the first pass bit for bit against an engine that steps."""
    c = CASES[case]
    mod, seed = c['mod'], c['seed']
    cfg = fm.EnvConfig(**c['kw'])
    ocfg = mod.Config(**{k: getattr(cfg, k) for k in mod.Config.__dataclass_fields__})
    n, N, T = c.get('n', N_ENVS), cfg.N, cfg.episode_length
    sample = np.unique(np.concatenate([np.arange(0, n, c['stride']), [n - 1]]))
    import gc
    gc.collect()
    torch.cuda.empty_cache()   # the ring takes 208 GB at cfg 3: nothing of an earlier test may linger in the allocator's cache
    eng = fm.RolloutEngine(cfg, n, device=DEV, seed=seed, tune_placement=0)
    ref_eng = fm.RolloutEngine(cfg, n, device=DEV, seed=seed, tune_placement=0)
    ring = fm.OutputRing(eng, T)
    orc = make_oracle(mod, ocfg, sample, seed)
    s = torch.as_tensor(sample, device=DEV)
    eng.reset(); ref_eng.reset(); orc.reset()
    gen = torch.Generator(device=DEV); gen.manual_seed(11)
    for rnd in range(2):
        tape = torch.randint(0, 5, (T, n, N), device=DEV, generator=gen, dtype=torch.int32)
        c0 = eng.launch_counts()[0]
        eng.rollout(tape, mode='span', ring=ring)
        assert eng.launch_counts()[0] - c0 == T and eng.phase == (-1 if case == 'fnav' else 0)
        assert eng.outs is ring.sets[T - 1] and eng.obs.data_ptr() == ring.obs[T - 1].data_ptr()   # the current set = the LAST step's slot
        host_tape = tape.cpu().numpy()
        for t in range(T):
            ref = orc.step(host_tape[t][sample])
            got = (ring.obs[t][s], None, ring.node_obs[t][s], ring.adj_env[t][s], ring.reward[t][s], ring.done[t][s],
                   ring.info_planes[t].permute(1, 2, 0)[s])
            check(mod, tuple(x.cpu().numpy() if x is not None else None for x in got), ref, '%s pass %d slot %d' % (case, rnd, t))
            if rnd == 0:   # the same steps one launch each, one output set: slot t == that set after step t
                ref_eng.step(tape[t])
                for k, slot in (('obs', ring.obs[t]), ('node_obs', ring.node_obs[t]), ('adj_env', ring.adj_env[t]), ('reward', ring.reward[t]),
                                ('done', ring.done[t])):
                    assert torch.equal(getattr(ref_eng, k), slot), (k, t)
                assert torch.equal(ref_eng.outs.info_planes, ring.info_planes[t]), ('info', t)
    eng.close(); ref_eng.close()
    del ring, eng, ref_eng
    gc.collect()
    torch.cuda.empty_cache()
