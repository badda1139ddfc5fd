"""measurement aid (GPU box): bench.py's four `secondary` entries of nav_fairassign_fairrew_formation_graph (one launch per step / span,
lockstep start / episodes ending at all phases), twice; FMARL_LIB selects a library variant.
usage: python tools/fnav_lines.py [config=fnav] [modes=eager,span,steady,steady-span] [repeats=2]"""
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
name = sys.argv[1] if len(sys.argv) > 1 else 'fnav'
import re
if name not in bench.CONFIGS and re.fullmatch(r'fnav\d+', name):   # fnav<N>: the scenario at N agents + N goals + 3 obstacles
    k = int(name[4:])
    bench.CONFIGS[name] = dict(workload='nav_fairassign_fairrew_formation_graph, %d agents + %d goals + 3 obstacles, %%d envs per GPU' % (k, k),
                               env=dict(bench.CONFIGS['fnav']['env'], num_agents=k, num_landmarks=k), n_envs=65536, cpu_envs=16, cpu_episodes=2)
modes = sys.argv[2].split(',') if len(sys.argv) > 2 else ['eager', 'span', 'steady', 'steady-span']
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    for mode in modes:
        d = bench.secondary_line(name, mode, dev)
        print(name, mode, 'ms_per_step %.4f kernel %.4f frac %.3f launches %d kernel=%s' % (d['ms_per_step'], d['kernel_avg_ms'], d['frac'], d['kernel_launches'], d['kernel']), flush=True)
