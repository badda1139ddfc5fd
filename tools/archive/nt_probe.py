"""measurement aid (GPU box): plain against non-temporal 16-byte stores in the pure store streams (fmarl_store_stream shapes 1 / 2 against
3 / 4), scattered chunk order, one step's byte count of cfg 3.  Round 4: non-temporal is 5-10 % SLOWER in every shape; the emission keeps
plain stores.  usage: python tools/archive/nt_probe.py"""
import ctypes as C, sys, torch, math
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fair_marl_amd import _lib
lib=_lib.load(); dev=torch.device('cuda:0')
nbytes=8317*1000*1000//16*16
buf=torch.empty(nbytes,dtype=torch.uint8,device=dev)
st=C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def order(chunks):
    o=int(chunks*0.6180339887)|1
    while math.gcd(o,chunks)!=1: o+=2
    return o
for rnd in range(2):
  for shape in (1,3,2,4):
    for chunk in (1<<16,1<<20):
        chunks=(nbytes//16+chunk//16-1)//(chunk//16)
        o=order(chunks)
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        call=lambda: _lib.check(lib.fmarl_store_stream(buf.data_ptr(),nbytes,shape,chunk,o,0,st),'x')
        call(); e0.record()
        for _ in range(8): call()
        e1.record(); e1.synchronize()
        ms=e0.elapsed_time(e1)/8
        print('shape %d chunk %8d scattered: %.4f ms %.3f TB/s'%(shape,chunk,ms,nbytes/ms/1e9),flush=True)
