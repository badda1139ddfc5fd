import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import fair_marl_amd as fm
cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
n = 65536
eng = fm.RolloutEngine(cfg, n, device='cuda:0', seed=1, async_reset=False)
tape = torch.randint(0, 5, (32, n, 32), device='cuda:0', dtype=torch.int32)
eng.reset()
K = 4000
eng.profile_enable(K)
import time
t0 = time.perf_counter()
for t in range(K):
    eng.step(tape[t % 32])
torch.cuda.synchronize()
print('wall per step %.3f ms' % ((time.perf_counter() - t0) / K * 1e3))
ms = np.array(eng.profile_read())
full = ms[ms > 1.0]
for i in range(0, len(full), 240):
    c = full[i:i + 240]
    print('%5d  mean %.3f  min %.3f  max %.3f' % (i, c.mean(), c.min(), c.max()))
