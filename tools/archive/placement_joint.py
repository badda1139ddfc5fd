"""measurement aid: is a (node_obs, adj) pair carved out of ONE allocation always a fast pair?
Times the emission-only kernel (fmarl_rebuild_graph) at BASELINE config 3 for k joint allocations (node_obs followed by adj,
adj at a 2 MiB aligned offset) and for k x k separately allocated pairs.  usage (GPU box): python tools/placement_joint.py [k]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import fair_marl_amd as fm

k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
n, dev = 65536, 'cuda:0'
eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, async_reset=False, tune_placement=0)
obs = torch.zeros(n, 32, 7, device=dev)
rec = torch.zeros(n, eng.episode_record_words, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(node, adj, reps=3):
    eng.rebuild_graph(obs, rec, node_obs=node, adj_env=adj)
    e0.record()
    for _ in range(reps):
        eng.rebuild_graph(obs, rec, node_obs=node, adj_env=adj)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


nn, na = eng.node_obs.numel(), eng.adj_env.numel()
align = (2 << 20) // 4
off = (nn + align - 1) // align * align
joint = []
for _ in range(k):
    buf = torch.empty(off + na, dtype=torch.float32, device=dev)
    joint.append((buf, buf[:nn].view_as(eng.node_obs), buf[off:off + na].view_as(eng.adj_env)))
print('addresses:', ' '.join('%x/%x' % (x.data_ptr(), a.data_ptr()) for _, x, a in joint), flush=True)
print('joint allocations (node_obs | adj in one buffer):', ' '.join('%.3f' % t(x, a) for _, x, a in joint), flush=True)
# cross pairs between the joint buffers: node of i with adj of j
T = np.array([[t(joint[i][1], joint[j][2]) for j in range(k)] for i in range(k)])
np.set_printoptions(precision=3, linewidth=200)
print('node of buffer i with adj of buffer j:'); print(T)
nodes = [torch.empty_like(eng.node_obs) for _ in range(2)]
adjs = [torch.empty_like(eng.adj_env) for _ in range(4)]
print('separate allocations:'); print(np.array([[t(x, a) for a in adjs] for x in nodes]))
