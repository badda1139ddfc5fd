"""measurement aid (GPU box): what bounds the generic emission path (navigation_graph at 10 agents, E = 23: 1 012-byte ego rows)?

Same process, same ring of time slots (bench.py's n10 config, 65 536 envs):
  1. the span kernel itself (step_span_kernel over an episode's slots), per step;
  2. fmarl_store_pattern: the kernel's OWN store pattern with everything but the stores removed -- 2 816-byte windows (64 rows of 44
     bytes) at 4-byte aligned starts as aligned 16-byte chunks + edge dwords, the adjacency as one dword per lane and store, a workgroup
     per group of envs walking the slots in order -- in dispatch order and scattered like env_block;
  3. the same bytes as plain 16-byte streams (fmarl_store_stream: bench.py's store ceiling);
  4. the pattern with the windows padded to 64-byte multiples and aligned starts (what padding ROW GROUPS would give at best).
usage: python tools/n10_pattern.py [config=n10] [reps=5]"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import fair_marl_amd as fm  # noqa: E402
from fair_marl_amd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'n10'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
spec = bench.CONFIGS[name]
cfg = fm.EnvConfig(**spec['env'])
n, dev, ep = spec['n_envs'], torch.device('cuda:0'), cfg.episode_length
torch.cuda.set_device(dev)
lib = _lib.load()
N, E, F = cfg.N, cfg.E, cfg.node_feat
eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, tune_placement=0)
ring = fm.OutputRing(eng, ep)
epb = eng.envs_per_workgroup
g = torch.Generator(device=dev); g.manual_seed(2000)
tape = torch.randint(0, 5, (ep, n, N), device=dev, generator=g, dtype=torch.int32)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
T = ep - 1


def ms_of(fn, reps=reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn()
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


# 1. the kernel
eng.reset()
eng.rollout(tape, mode='span', ring=ring)
torch.cuda.synchronize()
eng.profile_enable(4 * ep)
for _ in range(4):
    eng.rollout(tape, mode='span', ring=ring)
torch.cuda.synchronize()
ms, steps = eng.profile_read(with_steps=True)
span = [(m, s) for m, s in zip(ms, steps) if s > 1]
k_ms = min(m / s for m, s in span)
k_avg = sum(m for m, _ in span) / sum(s for _, s in span)
node_slot, adj_slot = n * N * E * F * 4, n * E * E * 4
step_bytes = bench.algorithmic_bytes(cfg) * n * N
print('%s: %d envs, %d envs per workgroup; a step writes node_obs %.1f MB + adj %.1f MB (+ obs / reward / done / info: %.1f MB); algorithmic %.1f MB'
      % (name, n, epb, node_slot / 1e6, adj_slot / 1e6, (step_bytes - node_slot - adj_slot) / 1e6, step_bytes / 1e6))
print('1. step_span_kernel (%d-step launches into time slots): %.4f ms per step (best launch %.4f) = %.2f TB/s of node_obs + adj'
      % (span[0][1], k_avg, k_ms, (node_slot + adj_slot) / k_avg / 1e9))

# 2. its store pattern alone
groups = (n + epb - 1) // epb


def coprime(groups):
    o = int(groups * 0.6180339887) | 1
    while math.gcd(o, groups) != 1:
        o += 2
    return o


def pattern(window, node_group, order, slots=T, node=ring.node_obs, node_slot_b=node_slot):
    _lib.check(lib.fmarl_store_pattern(node.data_ptr(), ring.adj_env.data_ptr(), node_group, epb * E * E * 4, groups, slots, node_slot_b, adj_slot,
                                       window, order, st), 'fmarl_store_pattern')


rows = {}
for label, order in (('dispatch order', 1), ('scattered', coprime(groups))):
    t = ms_of(lambda: pattern(64 * F * 4, epb * N * E * F * 4, order)) / T
    rows[label] = t
    print('2. store pattern, %d-byte windows at 4-byte aligned starts, dword adjacency, %s: %.4f ms per step = %.2f TB/s; the kernel is at %.3f of it'
          % (64 * F * 4, label, t, (node_slot + adj_slot) / t / 1e9, t / k_avg))
# 2b. one launch per step (slots = 1): the pattern's ceiling for fmarl_step
t1 = ms_of(lambda: pattern(64 * F * 4, epb * N * E * F * 4, coprime(groups), slots=1))
print('2b. the same, ONE slot per launch (a launch per step), scattered: %.4f ms' % t1)

# 3. plain 16-byte streams over the same bytes (bench.py's ceiling)
c = bench.store_ceiling(dev, step_bytes, T, dst=ring.node_obs)
print('3. plain 16-byte store streams over the node_obs ring: %.4f ms per step (%s); the kernel is at %.3f of it' % (c['ms_per_step'], c['shape'], c['ms_per_step'] / k_avg))

# 4. windows padded to 64-byte multiples at 64-byte aligned starts (what padding row groups could reach): 2 816 -> 2 816 is already 44 x 64;
# aligned starts need the GROUP to be a multiple of 64 bytes: pad the group
ng = epb * N * E * F * 4
ng64 = (ng + 63) // 64 * 64
if groups * ng64 <= ring.node_obs[0].numel() * 4 * 1.01 or True:
    slot_b = min(groups * ng64, node_slot // 64 * 64)
    t = ms_of(lambda: pattern(64 * F * 4 // 64 * 64 or 64, ng64, coprime(groups), node_slot_b=slot_b)) / T
    print('4. the same windows (%d bytes) with every group starting on a 64-byte boundary (group %d -> %d bytes): %.4f ms per step = %.2f TB/s'
          % (64 * F * 4 // 64 * 64, ng, ng64, t, (slot_b + adj_slot) / t / 1e9))
print('closure: kernel / pattern (scattered) = %.3f, pattern / plain streams = %.3f' % (rows['scattered'] / k_avg, c['ms_per_step'] / rows['scattered']))
