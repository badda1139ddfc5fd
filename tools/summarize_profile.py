#!/usr/bin/env python
"""Summarise gpurun_out/<tag>/ (written by tools/profile.sh on the GPU box) into profiles/:
  profiles/<tag>_<cfg>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_<cfg>_summary.md         bench line + per-kernel table + HBM traffic from the PMC passes
  profiles/pmc_traffic.json               per-config HBM bytes per step_kernel launch (read by bench.py)
HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE
are collected in separate --pmc passes, are in KiB, and FETCH_SIZE is doubled on gfx950 (it counts
128-byte requests at 64 bytes)."""
import csv
import glob
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cfg = (sys.argv + ['r1', 'cfg3'])[1:3]
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')
stem = tag if tag.endswith('_' + cfg) else '%s_%s' % (tag, cfg)
os.makedirs(dst, exist_ok=True)


def one(pattern):
    return max(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)   # gpurun merges: older runs stay around


def is_kernel(full_name, k):
    """`full_name`: a demangled kernel name of the trace; `k`: 'step_kernel', 'formation_kernel<true>' (any further template arguments match)."""
    name = full_name.split('(')[0].split('::')[-1].strip()
    return name.split('<')[0] == k.split('<')[0] and name.startswith(k.rstrip('>'))


def counter_mean(path, kernel, counter):
    names = kernel if isinstance(kernel, tuple) else (kernel,)
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(path))
            if any(is_kernel(r['Kernel_Name'], k) for k in names) and r['Counter_Name'] == counter]
    return float(np.mean(vals)), float(np.max(vals)), len(vals)


bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])   # the compact stdout line
# (round 6: the line is compact; the full record of the same run -- bench_detail.json, copied by tools/profile.sh -- holds the rest)
detail = json.load(open(os.path.join(src, 'bench_detail.json'))) if os.path.exists(os.path.join(src, 'bench_detail.json')) else bench
scen = 'fnav' if cfg.startswith('fnav') else cfg
stats = one('trace/*/*kernel_stats.csv')
shutil.copy(stats, os.path.join(dst, '%s_kernel_stats.csv' % stem))
mode = bench['config'].get('launch_mode', 'step')
small = 'small' in bench['roofline'].get('kernel', '')   # (small batches of navigation_graph run the step_small / step_span_small kernels)
if mode == 'span':   # the dominant kernel is the span kernel (a launch = a run of steps)
    kname = {'cfg4': 'formation_span_kernel', 'fnav': 'fairnav_span_kernel'}.get(scen, 'step_span_small_kernel' if small else 'step_span_kernel')
else:
    kname = {'cfg4': 'formation_kernel<true>', 'fnav': 'fairnav_kernel<true>'}.get(scen, ('step_small_kernel' if small else 'step_kernel', 'step_end_kernel'))   # the step launch: 24 + 1 per episode
f_mean, f_max, nf = counter_mean(one('pmc_fetch/*/*counter_collection.csv'), kname, 'FETCH_SIZE')
w_mean, w_max, nw = counter_mean(one('pmc_write/*/*counter_collection.csv'), kname, 'WRITE_SIZE')
traffic_mean = (2.0 * f_mean + w_mean) * 1024.0
traffic_full = (2.0 * f_max + w_max) * 1024.0
tpath = os.path.join(dst, 'pmc_traffic.json')
allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
spl = bench['roofline'].get('kernel_steps_per_launch', 1.0)
allt['%s/%s%s' % (cfg, mode, '-ring' if bench['roofline'].get('slots') == 'ring' else '')] = dict(n_envs=bench['config']['n_envs_per_gpu'], hbm_bytes_per_step=traffic_mean / spl, steps_per_launch=spl, hbm_bytes_per_launch=traffic_mean, hbm_bytes_full_launch=traffic_full, fetch_kib_mean=f_mean,
                 write_kib_mean=w_mean, launches=nf, source='%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE '
                 '(separate passes), %s rows, (2*FETCH_SIZE + WRITE_SIZE)*1024' % (tag, ' + '.join(kname) if isinstance(kname, tuple) else kname))
json.dump(allt, open(tpath, 'w'), indent=1, sort_keys=True)

rows = list(csv.DictReader(open(stats)))
out = os.path.join(dst, '%s_summary.md' % stem)
with open(out, 'w') as f:
    f.write('# %s %s: bench line, kernel trace, HBM counters\n\n' % (tag, cfg))
    f.write('Commands (tools/profile.sh, on the MI355X box): `python bench.py --config %s` (bench line); '
            '`rocprofv3 --kernel-trace --stats -- python3 bench.py --config %s --steps 200 --warmup 50 --no-cpu-baseline` '
            '(kernel table); `rocprofv3 --pmc FETCH_SIZE` and `rocprofv3 --pmc WRITE_SIZE` in separate passes over '
            '`python3 bench.py --config %s --steps 50 --warmup 25 --no-cpu-baseline` (traffic); all with `--no-secondary` '
            '(the other configs of the default line run the same kernel names).\n\n' % (cfg, cfg, cfg))
    f.write('## bench.py JSON line (stdout; the full record of the run is bench_detail.json)\n\n```json\n%s\n```\n\n' % json.dumps(bench, indent=1))
    f.write('## rocprofv3 kernel stats\n\n| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n')
    for r in rows[:8]:
        f.write('| %s | %s | %.1f | %.1f | %.1f | %s |\n' % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs']) / 1e3,
                                                       float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Percentage']))
    rl = detail['roofline']
    knames = kname if isinstance(kname, tuple) else (kname,)
    krows = [r for r in rows if any(is_kernel(r['Name'], k) for k in knames)]
    calls = sum(int(r['Calls']) for r in krows)
    krow = dict(AverageNs=sum(float(r['TotalDurationNs']) for r in krows) / calls, Calls=calls)
    kname = ' + '.join(knames)
    traced = [l for l in open(os.path.join(src, 'trace.log')) if l.startswith('{')]
    traced_ms = json.loads(traced[-1])['roofline']['kernel_avg_ms'] if traced else float('nan')
    f.write('\n%s average: rocprofv3 %.1f us over %s launches vs %.1f us measured live by bench.py with hipEvents '
            '(%d launches, un-profiled run).  Inside the traced run itself bench.py measured %.1f us over its 200 timed '
            'launches (the trace also contains the 50 warm-up launches; profiled runs clock lower, MI355X_MICROARCH.md DVFS note 2).\n'
            % (kname, float(krow['AverageNs']) / 1e3, krow['Calls'], rl['kernel_avg_ms'] * 1e3, rl['kernel_launches'], traced_ms * 1e3))
    if mode == 'span':
        # per-launch durations of the span kernel from the kernel trace (a launch = a run of steps; --stats averages the
        # warm-up's launches in as well)
        def span_launches(pattern):
            rows_ = [r for r in csv.DictReader(open(one(pattern))) if is_kernel(r['Kernel_Name'], knames[0])]
            return [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows_]
        d = span_launches('trace/*/*kernel_trace.csv')
        f.write('\nSpan launches of the traced run (ms each, in order; the bench line above reports %.1f steps per launch and measured '
                '%.3f ms per launch live): %s\n' % (rl.get('kernel_steps_per_launch', 0), rl['kernel_avg_ms'], ', '.join('%.3f' % v for v in d)))
        if glob.glob(os.path.join(src, 'trace_driver/*/*kernel_trace.csv')):
            dd = span_launches('trace_driver/*/*kernel_trace.csv')
            line = [l for l in open(os.path.join(src, 'trace_driver.log')) if l.startswith('{')]
            drl = json.loads(line[-1])['roofline'] if line else {}
            f.write('\nThe driver\'s command, `python3 bench.py --gpus 1 --steps 20 --warmup 5`, under the same trace: span launches %s ms '
                    '(the warm-up\'s run of 4 steps, then the timed run of 19); that run\'s own line: kernel_avg_ms %.3f over %.0f steps per '
                    'launch, frac %.3f.\n' % (', '.join('%.3f' % v for v in dd), drl.get('kernel_avg_ms', float('nan')),
                                             drl.get('kernel_steps_per_launch', float('nan')), drl.get('frac', float('nan'))))
    f.write('\n## HBM traffic of %s (PMC)\n\n' % kname)
    f.write('| counter | mean KiB / launch | max KiB / launch | launches |\n|---|---|---|---|\n')
    f.write('| FETCH_SIZE | %.0f | %.0f | %d |\n| WRITE_SIZE | %.0f | %.0f | %d |\n\n' % (f_mean, f_max, nf, w_mean, w_max, nw))
    f.write('HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 = **%.3f GB** mean (%.3f GB for a full, '
            'non-episode-end launch) against %.3f GB algorithmic per launch -> ratio %.3f.\n'
            % (traffic_mean / 1e9, traffic_full / 1e9, rl['algorithmic_bytes_per_launch'] / 1e9,
               traffic_mean / rl['algorithmic_bytes_per_launch']))
print(open(out).read())
