"""measurement aid: step time of a bench config with node_obs / adj emission switched off one at a time
(FmarlOutputs pointers set to NULL), to see which of the two streams a change to the emission code moves.
usage (GPU box): python tools/emit_split.py <config> [steps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import fair_marl_amd as fm
from fair_marl_amd import _lib
from bench import CONFIGS


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'n10'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    c = CONFIGS[name]
    cfg = fm.EnvConfig(**c['env'])
    dev = torch.device('cuda:0')
    eng = fm.RolloutEngine(cfg, c['n_envs'], device=dev, seed=1, async_reset=False, tune_placement=0)
    tape = torch.randint(0, 5, (32, c['n_envs'], cfg.N), device=dev, dtype=torch.int32)
    base = eng.outs
    variants = (('node + adj', True, True), ('node only', True, False), ('adj only', False, True), ('neither', False, False))
    if os.environ.get('EMIT_SPLIT_ONLY'):   # one variant (for a PMC pass over the process)
        variants = (variants[int(os.environ['EMIT_SPLIT_ONLY'])],)
    for label, node, adj in variants:
        o = eng.new_output_set()
        o.c = _lib.FmarlOutputs(o.obs.data_ptr(), o.node_obs.data_ptr() if node else None, o.adj_env.data_ptr() if adj else None,
                                o.reward.data_ptr(), o.done.data_ptr(), o.info_planes.data_ptr() if o.info_planes is not None else None,
                                None, None)
        eng.use_outputs(o)
        eng.reset()
        for t in range(25):
            eng.step(tape[t % 32], auto_reset=True)
        torch.cuda.synchronize()
        eng.profile_enable(steps)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(steps):
            eng.step(tape[t % 32], auto_reset=True)
        e1.record()
        torch.cuda.synchronize()
        k = eng.profile_read()
        ks = sorted(k)
        k = sum(k) / max(len(k), 1)
        eng.profile_enable(0)
        print('%-11s ms_per_step=%.4f  kernel_avg_ms=%.4f  (median %.4f, max %.4f, launches over 2x median: %d)' % (
            label, e0.elapsed_time(e1) / steps, k, ks[len(ks) // 2], ks[-1], sum(1 for v in ks if v > 2 * ks[len(ks) // 2])), flush=True)
    eng.use_outputs(base)


if __name__ == '__main__':
    main()
