import sys, time, torch
sys.path.insert(0, '/root/repo')
import fair_marl_amd as fm
dev = 'cuda:0'
cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
n = 65536
eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, tune_placement=0)
t0 = time.perf_counter()
buf = fm.DeviceRolloutBuffer(eng)
torch.cuda.synchronize()
print('buffer of %d slots allocated + zeroed in %.2f s; free %.1f GB' % (buf.T + 1, time.perf_counter() - t0, torch.cuda.mem_get_info()[0] / 1e9), flush=True)
g = torch.Generator(device=dev); g.manual_seed(0)
tape = torch.randint(0, 5, (25, n, cfg.N), device=dev, generator=g, dtype=torch.int32)
buf.reset()
for rnd in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    buf.insert_span(tape[:24]); buf.insert_step(tape[24])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('episode as insert_span + insert_step: %.3f ms per step' % (dt / 25 * 1e3), flush=True)
    buf.after_update()
buf.reset()
for rnd in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(25):
        buf.insert_step(tape[t])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('episode as 25 insert_step calls (a policy in the loop): %.3f ms per step' % (dt / 25 * 1e3), flush=True)
    buf.after_update()
