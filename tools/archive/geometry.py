"""measurement aid (GPU box): workgroups / threads / LDS bytes / envs per workgroup of every bench config (fmarl_launch_geometry) and how
many of those workgroups a CU holds by LDS -- 161 280 usable bytes is what was measured (profiles/archive/r3_notes.md).  usage: python tools/archive/geometry.py"""
import sys; sys.path.insert(0,'/root/repo')
import bench, fair_marl_amd as fm
for name in ('cfg3','n10','cfg2','cfg4','fnav'):
    c=bench.CONFIGS[name]
    eng=fm.RolloutEngine(fm.EnvConfig(**c['env']), c['n_envs'], device='cuda:0', tune_placement=0)
    g=eng.launch_geometry(); print(name, g, 'WGs/CU by LDS at 161280 usable: %d' % (161280//g[2]), 'generations %.2f' % (g[0]/(256*min(161280//g[2], 2048//g[1]))))
