"""measurement aid: emission-only launch time for every (node_obs allocation, adj allocation) pair."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fair_marl_amd as fm
cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
n, dev, K = 65536, 'cuda:0', 6
eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, async_reset=False, tune_placement=0)
obs = torch.zeros(n, 32, 7, device=dev)
rec = torch.zeros(n, eng.episode_record_words, dtype=torch.int32, device=dev)
nodes = [eng.node_obs] + [torch.empty_like(eng.node_obs) for _ in range(K - 1)]
adjs = [eng.adj_env] + [torch.empty_like(eng.adj_env) for _ in range(K - 1)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(**kw):
    for _ in range(2): eng.rebuild_graph(obs, rec, **kw)
    e0.record()
    for _ in range(5): eng.rebuild_graph(obs, rec, **kw)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 5
print('node only:', ' '.join('%.3f' % t(node_obs=x, want_adj=False) for x in nodes))
print('adj only :', ' '.join('%.3f' % t(adj_env=x, want_node_obs=False) for x in adjs))
print('node addr:', ' '.join('%x' % x.data_ptr() for x in nodes))
print('adj addr :', ' '.join('%x' % x.data_ptr() for x in adjs))
for i, x in enumerate(nodes):
    print('node %d + adj j:' % i, ' '.join('%.3f' % t(node_obs=x, adj_env=a) for a in adjs))
