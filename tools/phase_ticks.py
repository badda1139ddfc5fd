"""measurement aid (GPU box): where a wave of formation_kernel<true> / fairnav_kernel<true> spends its cycles -- the FMARL_TICK sites of a
-DFMARL_MEASURE build (tools/mkvariant.sh measure -DFMARL_MEASURE), summed over the waves of one launch under full load.
usage: FMARL_LIB=fair_marl_amd/csrc/variants/libfmarl_measure.so python tools/phase_ticks.py [config]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import fair_marl_amd as fm  # noqa: E402
from fair_marl_amd import _lib  # noqa: E402

NAMES_NAV = ['loads+tables+barrier', 'physics', 'agent rows', 'scan statistics', 'stats+hits+reward+stores', 'node_obs',
             'adj (odd workgroups: first)', 'adj (even workgroups: last)', 'wait at the barrier behind the physics', 'wait at the barrier behind the agent rows',
             'wait at the emission\'s first barrier (generic rows)']
NAMES_FNAV = ['loads+tables+barrier', 'physics', 'distance table', 'assignment', 'status+bookkeeping', 'walk', 'reward+state+info (before the walk)',
              'obs+occupancy state+record', 'node rows', 'in-kernel reset of the ended envs: the rest', 'adj', 'in-kernel reset: the barrier that finds ended envs + the pre-draw of their Philox blocks',
              'in-kernel reset: the placement (teams of lanes; round 5: the first lane of an ended env)', 'in-kernel reset: the barrier behind the placement']
NAMES = ['loads+tables+barrier', 'physics', 'keys+ring+slots', 'agent x slot distances', 'occupancy', 'matchings', 'sets+walk',
         'obs+record', 'stats+hits+reward', 'state stores', 'info planes', 'node rows', 'adj']


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
    kw, n = dict(bench.CONFIGS[name]['env']), bench.CONFIGS[name]['n_envs']
    if len(sys.argv) > 2:   # fnav with episodes ending at all phases: python tools/phase_ticks.py fnav 0.5 600  (then the table shows
        kw['min_dist_thresh'] = float(sys.argv[2])   # the RESET pass of the last launch: it overwrites the step pass's rows)
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    cfg = fm.EnvConfig(**kw)
    eng = fm.RolloutEngine(cfg, n, device='cuda:0', seed=5, tune_placement=0)
    gen = torch.Generator(device='cuda:0'); gen.manual_seed(2)
    eng.reset()
    tape = torch.randint(0, 5, (25, n, cfg.N), device='cuda:0', generator=gen, dtype=torch.int32)
    if os.environ.get('FMARL_TICKS_SPAN'):   # the LAST step of one span launch of FMARL_TICKS_SPAN steps (every step rewrites the wave's row)
        eng.step_span(tape[:int(os.environ['FMARL_TICKS_SPAN'])].contiguous())
        steps = 0
    for t in range(steps):
        eng.step(tape[t % 25])
    lib = _lib.load()
    if name == 'cfg4' and os.environ.get('FMARL_HSTAT'):   # (a -DFMARL_MEASURE -DFMARL_HSTAT build) the slot matchings of the last step
        hs = (C.c_ulonglong * 8)()
        lib.fmarl_measure_hstat.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
        torch.cuda.synchronize()
        assert lib.fmarl_measure_hstat(hs, 1) == 0
        eng.step(tape[steps % 25])
        assert lib.fmarl_measure_hstat(hs, 1) == 0
        for w, label in ((0, 'current slots'), (1, 'previous slots')):
            run, skip, rows, its = hs[4 * w:4 * w + 4]
            print('matching on the %s: %d run, %d skipped; per run %.2f rows through the augmenting search, %.2f search iterations'
                  % (label, run, skip, rows / max(1, run), its / max(1, run)))
    waves = -(-n // eng.envs_per_workgroup) * 4
    out = (C.c_double * 16)()
    lib.fmarl_measure_ticks.argtypes = [C.POINTER(C.c_double), C.c_int]
    assert lib.fmarl_measure_ticks(out, waves) == 0
    tot = sum(out[:14])
    print('%s: %d waves, %.0f cycles per wave' % (name, waves, tot / waves))
    for k, nm in enumerate(NAMES_FNAV if name.startswith('fnav') else (NAMES if name == 'cfg4' else NAMES_NAV)):
        print('  %-26s %8.0f cycles  %5.1f %%' % (nm, out[k] / waves, 100 * out[k] / tot))
    import numpy as np
    rows = np.zeros((waves, 16), dtype=np.uint32)
    lib.fmarl_measure_rows.argtypes = [C.c_void_p, C.c_int]
    assert lib.fmarl_measure_rows(rows.ctypes.data, waves) == 0
    if name not in ('fnav', 'fnav10', 'cfg4') and eng.envs_per_workgroup * cfg.N <= 64:   # small batches (step_body SMALL): wave 0 = the agents, waves 1 .. 3 = the emission
        for label, sel in (('wave 0 of a workgroup (agents)', rows[0::4]), ('waves 1 .. 3 (emission)', np.concatenate([rows[1::4], rows[2::4], rows[3::4]]))):
            print('%s: %.0f cycles per wave' % (label, sel[:, :14].sum() / len(sel)))
            for k, nm in enumerate(NAMES_NAV):
                print('    %-26s %8.0f cycles' % (nm, sel[:, k].mean()))
    start, end = rows[:, 15].astype(np.int64), rows[:, 14].astype(np.int64)
    t0 = start.min()
    print('wave starts (us after the first, 100 MHz clock): percentiles 10/50/90/99/100 = %s' % np.round(np.percentile((start - t0) / 100.0, [10, 50, 90, 99, 100]), 1))
    print('wave ends: percentiles 10/50/90/100 = %s;  lifetime mean %.1f us' % (np.round(np.percentile((end - t0) / 100.0, [10, 50, 90, 100]), 1), ((end - start) / 100.0).mean()))
    h, _ = np.histogram((start - t0) / 100.0, bins=12)
    print('start histogram:', h.tolist())


if __name__ == '__main__':
    main()
