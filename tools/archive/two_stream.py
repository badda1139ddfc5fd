"""measurement aid: what overlapping consecutive step kernels is worth.  The env batch of a bench config is split into k
sub-batches (RolloutEngine instances with env_offset, so the union is the same set of envs), each stepped on its own
stream; steps of one sub-batch are ordered by its stream, sub-batches are independent, so the tail of one kernel overlaps
the head of another sub-batch's next one.  usage (GPU box): python tools/two_stream.py <config> [k ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import fair_marl_amd as fm
from bench import CONFIGS


def run(name, k, steps=200):
    c = CONFIGS[name]
    cfg = fm.EnvConfig(**c['env'])
    dev = torch.device('cuda:0')
    n = c['n_envs'] // k
    streams = [torch.cuda.Stream(dev) for _ in range(k)] if k > 1 else [torch.cuda.current_stream(dev)]
    engs, tapes = [], []
    for j in range(k):
        with torch.cuda.stream(streams[j]):
            e = fm.RolloutEngine(cfg, n, device=dev, seed=1, env_offset=j * n, async_reset=False, tune_placement=0)
            e.reset()
            engs.append(e)
            tapes.append(torch.randint(0, 5, (32, n, cfg.N), device=dev, dtype=torch.int32))
    def go(count):
        for t in range(count):
            for j in range(k):
                with torch.cuda.stream(streams[j]):
                    engs[j].step(tapes[j][t % 32], auto_reset=True)
    go(25)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print('%s  sub-batches=%d x %d envs  ms_per_step(all envs)=%.4f  agent-steps/s=%.3e' % (name, k, n, dt * 1e3, c['n_envs'] * cfg.N / dt), flush=True)


if __name__ == '__main__':
    name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
    for k in [int(x) for x in sys.argv[2:]] or [1, 2, 4]:
        run(name, k)
