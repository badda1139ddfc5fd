import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import fair_marl_amd as fm
from oracle import formation_oracle as fo
from oracle.philox import PhiloxStream
case = int(sys.argv[1]) if len(sys.argv) > 1 else 41
rs = np.random.RandomState(1000 + case)
kind = case % 3
N = int(rs.randint(1, 10)); O = int(rs.randint(0, 5)); W = int(rs.randint(0, 3)); n = int(rs.choice([1, 2, 7, 33]))
ep = int(rs.choice([1, 2, 5, 9]))
kw = dict(num_agents=N, num_obstacles=O, episode_length=ep, max_speed=None if case % 5 == 4 else float(rs.choice([0.7, 2.0])),
          min_dist_thresh=float(rs.choice([0.05, 0.3, 0.6])), goal_rew=float(rs.choice([5, 2.5])), collision_rew=float(rs.choice([5, 1.0])))
seed = 77 + case
cfg = fm.EnvConfig(scenario_name='fair_graph_formation', num_landmarks=int(rs.randint(1, 3)), **kw)
ocfg = fo.Config(**{k: getattr(cfg, k) for k in fo.Config.__dataclass_fields__})
orc = fo.OracleFormationVecEnv(ocfg, n, mode='subproc', streams=lambda e, ep_: PhiloxStream(seed, e, ep_))
eng = fm.RolloutEngine(cfg, n, device='cuda:0', seed=seed, async_reset=bool(case % 2))
eng.reset(); orc.reset()
FORM_INFO = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13]
for t in range(2 * ep + 1):
    a = rs.randint(0, 5, size=(n, N))
    res = eng.step(torch.as_tensor(a, device='cuda:0'))
    ref = orc.step(a)
    got = res[6].cpu().numpy()[..., FORM_INFO]; want = ref[6]
    bad = np.argwhere(~np.isclose(got, want, rtol=1e-5, atol=1e-5))
    print('step', t, 'N', N, 'n', n, 'bad', len(bad))
    for b in bad[:6]:
        e, ag, k = b
        print('  env %d agent %d plane %d got %.9g want %.9g | row got %s | want %s' % (e, ag, FORM_INFO[k], got[e, ag, k], want[e, ag, k], np.array2string(got[e, ag], precision=6), np.array2string(want[e, ag], precision=6)))
    if len(bad): break
