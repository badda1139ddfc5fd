import os, sys, time, argparse, numpy as np, torch
sys.path.insert(0, os.getcwd())
import fair_marl_amd as fm
args = argparse.Namespace(scenario_name='navigation_graph', num_agents=3, num_landmarks=3, num_obstacles=3, episode_length=25)
def fn(r):
    def init():
        e = fm.GraphMPEEnv(args); e.seed(1 + 1000 * r); return e
    return init
n = 4096
env = fm.GraphSubprocVecEnv([fn(r) for r in range(n)])
env.reset()
acts = np.eye(5)[np.random.randint(0, 5, size=(n, 3))]
for _ in range(5): env.step(acts)
T = {k: 0.0 for k in ('async', 'device', 'cat', 'd2h', 'split')}
K = 50
for _ in range(K):
    t0 = time.perf_counter(); env.step_async(acts); t1 = time.perf_counter()
    outs = env._step_device(); torch.cuda.synchronize(); t2 = time.perf_counter()
    obs, ids, node, adj, rew, done, info = outs
    tensors = (obs, node, adj, rew, done, info)
    flat = torch.cat([t.detach().to(torch.float64).reshape(-1) for t in tensors]); torch.cuda.synchronize(); t3 = time.perf_counter()
    host = env._staging[:flat.numel()]; host.copy_(flat, non_blocking=True); torch.cuda.synchronize(); t4 = time.perf_counter()
    arr, o, res = host.numpy(), 0, []
    for t in tensors:
        res.append(arr[o:o + t.numel()].reshape(tuple(t.shape)).copy()); o += t.numel()
    t5 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): T[k] += v
print({k: round(v / K * 1e3, 3) for k, v in T.items()}, 'ms; bytes', flat.numel() * 8 / 1e6, 'MB')
