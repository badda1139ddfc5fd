"""measurement aid: with node_obs / adj fixed (tuned pair), does the placement of the small outputs or the state matter?"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fair_marl_amd as fm
cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
n, dev = 65536, 'cuda:0'
eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, async_reset=False)
print('placement matrix min %.3f first %.3f' % (min(map(min, eng.placement_ms)), eng.placement_ms[0][0]))
tape = torch.randint(0, 5, (32, n, 32), device=dev, dtype=torch.int32)
eng.reset()
def timed(outs, steps=48):
    eng.use_outputs(outs)
    for t in range(8): eng.step(tape[t % 32], auto_reset=False)
    eng.profile_enable(steps)
    for t in range(steps): eng.step(tape[t % 32], auto_reset=False)
    torch.cuda.synchronize()
    return float(np.mean(eng.profile_read()))
print('default small outputs %.3f' % timed(eng.outs))
keep = []
for k in range(6):
    keep.append(torch.empty((k + 1) * (13 << 20), dtype=torch.uint8, device=dev))
    o = eng.new_output_set()
    keep.append(o)
    print('fresh small outputs %d  %.3f' % (k, timed(o)))
o = eng.new_output_set()
eng.emit_info = False
o2 = eng.new_output_set()
print('no info planes  %.3f' % timed(o2))
