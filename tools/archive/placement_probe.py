"""measurement aid: does the step kernel's duration depend on WHERE its output buffers were allocated?
One process, one engine, several node_obs / adj allocations (and virtual-address offsets inside one), timed in turn."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fair_marl_amd as fm
cfg = fm.EnvConfig(num_agents=32, num_landmarks=32, num_obstacles=8)
n = 65536
dev = 'cuda:0'
eng = fm.RolloutEngine(cfg, n, device=dev, seed=1, async_reset=False)
tape = torch.randint(0, 5, (32, n, 32), device=dev, dtype=torch.int32)
eng.reset()

def timed(outs, steps=48):
    eng.use_outputs(outs)
    for t in range(8):
        eng.step(tape[t % 32], auto_reset=False)
    eng.profile_enable(steps)
    for t in range(steps):
        eng.step(tape[t % 32], auto_reset=False)
    torch.cuda.synchronize()
    return float(np.mean(eng.profile_read()))

E, F = cfg.E, cfg.node_feat
print('default set   %.3f ms  node@%x adj@%x' % (timed(eng.outs), eng.node_obs.data_ptr(), eng.adj_env.data_ptr()))
keep = []
for k in range(6):
    node = torch.empty(n, 32, E, F, dtype=torch.float32, device=dev)
    adj = torch.empty(n, E, E, dtype=torch.float32, device=dev)
    keep += [node, adj]
    o = eng.new_output_set(node_obs=node, adj_env=adj)
    print('fresh alloc %d %.3f ms  node@%x adj@%x' % (k, timed(o), node.data_ptr(), adj.data_ptr()))
# one slab, adj at different offsets behind node_obs
slab = torch.empty(n * 32 * E * F * 4 + n * E * E * 4 + (64 << 20), dtype=torch.uint8, device=dev)
nb = n * 32 * E * F * 4
for off in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, 17 << 20):
    node = slab[:nb].view(torch.float32).view(n, 32, E, F)
    adj = slab[nb + off: nb + off + n * E * E * 4].view(torch.float32).view(n, E, E)
    o = eng.new_output_set(node_obs=node, adj_env=adj)
    print('slab, adj offset %9d  %.3f ms' % (off, timed(o)))
keep2 = []
for k in range(6):
    pad = torch.empty((k + 1) * (37 << 20), dtype=torch.uint8, device=dev)   # perturb the allocator between slabs
    s2 = torch.empty(nb + n * E * E * 4, dtype=torch.uint8, device=dev)
    keep2 += [pad, s2]
    node = s2[:nb].view(torch.float32).view(n, 32, E, F)
    adj = s2[nb:].view(torch.float32).view(n, E, E)
    o = eng.new_output_set(node_obs=node, adj_env=adj)
    print('slab %d @%x  %.3f ms' % (k, s2.data_ptr(), timed(o)))
# adj in front of node_obs
s3 = torch.empty(nb + n * E * E * 4, dtype=torch.uint8, device=dev)
ab = n * E * E * 4
o = eng.new_output_set(node_obs=s3[ab:].view(torch.float32).view(n, 32, E, F), adj_env=s3[:ab].view(torch.float32).view(n, E, E))
print('slab, adj first  %.3f ms' % timed(o))
