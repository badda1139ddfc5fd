"""measurement aid: emission-only launch time with node_obs / adj allocated by hipExtMallocWithFlags
(default, contiguous, uncached, fine-grained) instead of torch's caching allocator."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fair_marl_amd as fm
from fair_marl_amd import _lib
hip = C.CDLL('libamdhip64.so')
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
n, dev = 65536, 'cuda:0'
eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, async_reset=False, tune_placement=0)
obs = torch.zeros(n, 32, 7, device=dev)
rec = torch.zeros(n, eng.episode_record_words, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
nb, ab = eng.node_obs.numel() * 4, eng.adj_env.numel() * 4
def timed(node_ptr, adj_ptr):
    def go():
        _lib.check(eng.lib.fmarl_rebuild_graph(eng.handle, obs.data_ptr(), rec.data_ptr(), n, C.c_void_p(node_ptr), C.c_void_p(adj_ptr), eng._stream()), 'rebuild')
    for _ in range(2): go()
    e0.record()
    for _ in range(5): go()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 5
print('torch first allocations     %.3f' % timed(eng.node_obs.data_ptr(), eng.adj_env.data_ptr()))
for name, flags in (('default', 0), ('contiguous', 4), ('uncached', 3), ('finegrained', 1), ('contiguous', 4), ('default', 0)):
    pn, pa = C.c_void_p(), C.c_void_p()
    r1 = hip.hipExtMallocWithFlags(C.byref(pn), nb, flags)
    r2 = hip.hipExtMallocWithFlags(C.byref(pa), ab, flags)
    if r1 or r2:
        print('%-12s alloc failed (%d, %d)' % (name, r1, r2)); continue
    print('hipExtMallocWithFlags %-12s %.3f   node@%x adj@%x' % (name, timed(pn.value, pa.value), pn.value, pa.value))
    torch.cuda.synchronize()
    hip.hipFree(pn); hip.hipFree(pa)
ps = C.c_void_p()
if hip.hipExtMallocWithFlags(C.byref(ps), nb + ab, 4) == 0:
    print('one contiguous slab          %.3f' % timed(ps.value, ps.value + nb))
    torch.cuda.synchronize(); hip.hipFree(ps)
