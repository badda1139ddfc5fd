"""measurement aid: lexifair_kernel alone (65 536 x 32 x 32 cdist-like costs), ms per launch"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fair_marl_amd as fm
cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
n, dev = 65536, 'cuda:0'
eng = fm.RolloutEngine(cfg, 64, device=dev, seed=1, tune_placement=0)
g = torch.Generator(device=dev); g.manual_seed(0)
a = torch.rand(n, 32, 2, device=dev, generator=g, dtype=torch.float64) * 2 - 1
b = (torch.rand(n, 32, 2, device=dev, generator=g, dtype=torch.float64) * 2 - 1) * 0.8
costs = torch.cdist(a, b).contiguous()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
perm = eng.lexifair(costs)
e0.record()
for _ in range(5): perm = eng.lexifair(costs)
e1.record(); e1.synchronize()
print('lexifair 65536 x 32: %.3f ms per launch, checksum %d' % (e0.elapsed_time(e1) / 5, int(perm.sum())))
