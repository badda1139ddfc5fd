"""measurement aid: what do the per-launch hipEvent pairs of fmarl_profile_enable cost per step?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fair_marl_amd as fm
cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
n, dev = 65536, 'cuda:0'
eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, async_reset=True)
tape = torch.randint(0, 5, (32, n, 32), device=dev, dtype=torch.int32)
eng.reset()
def run(K, prof):
    eng.profile_enable(K if prof else 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K): eng.step(tape[t % 32])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    if prof: eng.profile_read()
    return dt
run(50, False)
for rep in range(3):
    print('events on %.4f ms/step   events off %.4f ms/step' % (run(500, True), run(500, False)))
