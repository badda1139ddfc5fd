"""probe: device memory free after failed fmarl_ring_alloc attempts (more than the device has)"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from fair_marl_amd import _lib
lib = _lib.load()
torch.cuda.set_device(0)
torch.ones(4, device='cuda:0').sum().item()
free0, total = torch.cuda.mem_get_info()
print('free at start %d MiB' % (free0 >> 20))
slot = (total // (1 << 24) * (1 << 24)) * 3 // 4
stats = (C.c_uint64 * 8)()
for k in range(4):
    base, cookie = C.c_void_p(), C.c_void_p()
    rc = lib.fmarl_ring_alloc(slot, 2, 0, C.byref(base), C.byref(cookie))
    lib.fmarl_ring_stats(stats)
    print('attempt %d rc %d: free %d MiB (%+d MiB vs start); reserved %d MiB in %d ranges; %s'
          % (k, rc, torch.cuda.mem_get_info()[0] >> 20, (torch.cuda.mem_get_info()[0] - free0) >> 20, stats[0] >> 20, stats[2], lib.fmarl_last_error().decode()[:80]))
