"""measurement aid: do allocations from other regions of the 288 GB behave differently?  Spacer allocations of
growing size are held while a (node_obs, adj) pair is allocated and timed (emission-only launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fair_marl_amd as fm
cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
n, dev = 65536, 'cuda:0'
eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, async_reset=False, tune_placement=0)
obs = torch.zeros(n, 32, 7, device=dev)
rec = torch.zeros(n, eng.episode_record_words, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(**kw):
    for _ in range(2): eng.rebuild_graph(obs, rec, **kw)
    e0.record()
    for _ in range(5): eng.rebuild_graph(obs, rec, **kw)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 5
print('first allocations: pair %.3f node %.3f adj %.3f' % (t(node_obs=eng.node_obs, adj_env=eng.adj_env), t(node_obs=eng.node_obs, want_adj=False), t(adj_env=eng.adj_env, want_node_obs=False)))
GB = 1 << 30
for sp_n, sp_a in ((0, 0), (16, 0), (48, 0), (112, 0), (176, 0), (0, 16), (0, 48), (0, 112), (48, 48), (112, 48), (176, 16)):
    s1 = torch.empty(sp_n * GB, dtype=torch.uint8, device=dev) if sp_n else None
    node = torch.empty_like(eng.node_obs)
    s2 = torch.empty(sp_a * GB, dtype=torch.uint8, device=dev) if sp_a else None
    adj = torch.empty_like(eng.adj_env)
    print('spacer before node %3d GB, between node and adj %3d GB: pair %.3f  node %.3f  adj %.3f   (node@%x adj@%x)' % (
        sp_n, sp_a, t(node_obs=node, adj_env=adj), t(node_obs=node, want_adj=False), t(adj_env=adj, want_node_obs=False), node.data_ptr(), adj.data_ptr()))
    del s1, s2, node, adj
    torch.cuda.empty_cache()
