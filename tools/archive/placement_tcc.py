"""measurement aid: why is one (node_obs allocation, adj allocation) pair slower than another?

    cd /tmp && rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL \
        --output-format csv -d /tmp/ptcc -- python3 $REPO/tools/placement_tcc.py
    python3 $REPO/tools/placement_tcc.py --summarize /tmp/ptcc $REPO/gpurun_out/placement_tcc.log

Times the emission-only kernel (fmarl_rebuild_graph: writes node_obs + adj, touches no env state) for every pair of
3 x 6 allocations at BASELINE config 3, then launches the FASTEST and the SLOWEST pair five times each as the last ten
dispatches of the process (and says so on stdout), so that a PMC pass over the same run has the per-channel L2 -> fabric
write-request counters of both (16 TCC channels x 8 XCDs).
"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    import fair_marl_amd as fm
    cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
    n, dev = 65536, 'cuda:0'
    eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, async_reset=False, tune_placement=0)
    obs = torch.zeros(n, 32, 7, device=dev)
    rec = torch.zeros(n, eng.episode_record_words, dtype=torch.int32, device=dev)
    nodes = [eng.node_obs] + [torch.empty_like(eng.node_obs) for _ in range(2)]
    adjs = [eng.adj_env] + [torch.empty_like(eng.adj_env) for _ in range(5)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def t(reps, **kw):
        e0.record()
        for _ in range(reps):
            eng.rebuild_graph(obs, rec, **kw)
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / reps
    times = {}
    for i, x in enumerate(nodes):
        for j, a in enumerate(adjs):
            t(1, node_obs=x, adj_env=a)
            times[i, j] = t(3, node_obs=x, adj_env=a)
    best, worst = min(times, key=times.get), max(times, key=times.get)
    print('PAIR_TIMES ' + ' '.join('%d,%d=%.3f' % (i, j, v) for (i, j), v in sorted(times.items())))
    print('NODE_ADDR ' + ' '.join('%x' % x.data_ptr() for x in nodes))
    print('ADJ_ADDR ' + ' '.join('%x' % x.data_ptr() for x in adjs))
    torch.cuda.synchronize()
    tb = t(5, node_obs=nodes[best[0]], adj_env=adjs[best[1]])
    tw = t(5, node_obs=nodes[worst[0]], adj_env=adjs[worst[1]])
    print('LAST_TEN best=%s %.3f ms  worst=%s %.3f ms  (5 launches each, in this order)' % (best, tb, worst, tw), flush=True)


def summarize(prof_dir, log):
    import numpy as np
    rows = list(csv.DictReader(open(glob.glob(os.path.join(prof_dir, '*', '*counter_collection.csv'))[0])))
    disp = collections.OrderedDict()
    for r in rows:
        if 'rebuild_graph_kernel' not in r['Kernel_Name']:
            continue
        disp.setdefault(int(r['Dispatch_Id']), collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
    ids = sorted(disp)[-10:]
    groups = {'fastest pair': ids[:5], 'slowest pair': ids[5:]}
    if len(sys.argv) > 4:   # scalar counters (extra-counter YAML passes): one line per counter, fastest vs slowest pair
        print('\n`%s`:\n' % sys.argv[4])
        for l in open(log):
            if l.startswith('LAST_TEN'):
                print('    ' + l.strip())
        print('\n| counter | fastest pair | slowest pair | slowest / fastest |\n|---|---|---|---|')
        for c in sorted(disp[ids[0]]):
            f = np.mean([disp[d][c][0] for d in groups['fastest pair']]); w = np.mean([disp[d][c][0] for d in groups['slowest pair']])
            print('| %s | %.5g | %.5g | %.3f |' % (c, f, w, w / max(f, 1e-9)))
        return
    print('# (node_obs, adj) placement: per-channel L2 -> fabric write requests of the emission-only kernel\n')
    for l in open(log):
        if l.startswith(('LAST_TEN', 'PAIR_TIMES')):
            print('    ' + l.strip())
    print('\n128 TCC channels (16 per XCD x 8 XCDs); one row per counter, statistics over the channels, mean of 5 launches.\n')
    print('| pair | counter | sum | mean / channel | min | max | max / mean | std / mean |')
    print('|---|---|---|---|---|---|---|---|')
    for name, dids in groups.items():
        for c in sorted(disp[dids[0]]):
            v = np.mean([np.array(disp[d][c]) for d in dids], axis=0)
            print('| %s | %s | %.4g | %.4g | %.4g | %.4g | %.3f | %.3f |' % (name, c, v.sum(), v.mean(), v.min(), v.max(),
                                                                              v.max() / max(v.mean(), 1e-9), v.std() / max(v.mean(), 1e-9)))
    a = np.mean([np.array(disp[d]['TCC_EA0_WRREQ']) for d in groups['fastest pair']], axis=0)
    b = np.mean([np.array(disp[d]['TCC_EA0_WRREQ']) for d in groups['slowest pair']], axis=0)
    print('\nper-channel write requests, slowest / fastest pair: min %.3f max %.3f (same bytes, same kernel: only the page placement differs)'
          % ((b / np.maximum(a, 1)).min(), (b / np.maximum(a, 1)).max()))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--summarize':
        summarize(sys.argv[2], sys.argv[3])
    else:
        run()
