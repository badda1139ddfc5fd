"""Two (or more) ranks, each stepping its shard of the envs on the GPU, records gathered to rank 0 as bench.py does.

Run under ``python -m torch.distributed.run --nproc-per-node W tests/dist_rollout_check.py`` (started by
tests/test_hip_parity.py::test_two_ranks_gather_reproduces_the_unsharded_rollout).  The ranks share the box's
GPU(s) and exchange over gloo: RCCL refuses two ranks on one device, and the GPU box has one.  What is checked is
everything around the collective -- shard offsets, lockstep episode detection, record rotation, the learner-side
rebuild -- against ONE unsharded engine run on rank 0:
  * gathered obs / reward / done of every step == the unsharded engine's rows of that shard, bit for bit;
  * node_obs / adj rebuilt on rank 0 from the gathered obs + the once-per-episode record == the unsharded
    engine's node_obs / adj, bit for bit.
Prints ``DIST_CHECK_OK steps=<T> world=<W>`` on success (rank 0), raises otherwise.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import fair_marl_amd as fm  # noqa: E402
from fair_marl_amd.sharding import SpanGather, TrajectoryGather, shard_range  # noqa: E402


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    torch.cuda.set_device(device)
    dist.init_process_group('gloo')
    if 'formation' in sys.argv[1:]:   # BASELINE config 4's scenario: the node features need the per-step graph record too
        cfg = fm.EnvConfig(scenario_name='fair_graph_formation', num_agents=5, num_landmarks=1, num_obstacles=2, episode_length=6,
                           min_dist_thresh=0.3)
    elif 'fairnav' in sys.argv[1:]:   # episodes end env by env: the episode record travels with every step
        cfg = fm.EnvConfig(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=3, num_landmarks=3, num_obstacles=2,
                           episode_length=6, min_dist_thresh=0.35)
    else:
        cfg = fm.EnvConfig(num_agents=4, num_landmarks=4, num_obstacles=3, num_walls=1, episode_length=6)
    per, T, seed = 96, 20, 11                      # three auto-resets inside T steps
    n_total = per * world
    lo, hi = shard_range(n_total, world, rank)
    assert hi - lo == per
    eng = fm.RolloutEngine(cfg, per, device=device, seed=seed, env_offset=lo, emit_graph_record=True)
    tg = TrajectoryGather(per, cfg.N, cfg.obs_dim, device, dst=0, depth=2, episode_words=eng.episode_record_words,
                          graph_words=eng.step_record_words)
    sets = [eng.new_output_set(obs=r.obs, reward=r.reward, done=r.done, graph_record=r.graph) for r in tg.records]
    g = torch.Generator(device='cpu')
    g.manual_seed(5)
    tape = torch.randint(0, 5, (T, n_total, cfg.N), generator=g, dtype=torch.int32).to(device)   # same on every rank

    full = ref_sets = None
    if rank == 0:                                  # the oracle of this test: the same envs in ONE engine
        full = fm.RolloutEngine(cfg, n_total, device=device, seed=seed, env_offset=0)
        full.reset()
    if 'span' in sys.argv[1:]:
        return span_main(cfg, eng, full, tape, per, T, n_total, lo, hi, rank, world, device)
    eng.reset()
    assert eng.episode_started
    eng.pack_episode(out=tg.episode_record())
    tg.submit_episode()
    checked = 0
    for t in range(T):
        tg.record(t)
        eng.use_outputs(sets[t % 2])
        eng.step(tape[t, lo:hi], auto_reset=True)
        tg.submit(t)
        started = eng.episode_started
        if rank == 0:
            f_obs, _, f_node, f_adj, f_rew, f_done, _ = full.step(tape[t], auto_reset=True)
            assert full.episode_started == started
        if started:
            eng.pack_episode(out=tg.episode_record())
            tg.submit_episode()
        tg.pending[t % 2].wait()
        if rank == 0:
            torch.cuda.synchronize(device)
            episode = tg.gathered_episode()
            graphs = tg.gathered_graph(t)
            for r, (obs, rew, done) in enumerate(tg.gathered(t)):
                l2, h2 = shard_range(n_total, world, r)
                assert torch.equal(obs, f_obs[l2:h2]), 'obs of rank %d at step %d' % (r, t)
                assert torch.equal(rew, f_rew[l2:h2]) and torch.equal(done.bool(), f_done[l2:h2].bool())
                # node_obs / adj never travel: rebuilt from the gathered rows + this episode's record
                node, adj = eng.rebuild_graph(obs, episode[r], step_record=graphs[r])
                assert torch.equal(node, f_node[l2:h2]), 'rebuilt node_obs of rank %d at step %d' % (r, t)
                assert torch.equal(adj, f_adj[l2:h2, 0]), 'rebuilt adj of rank %d at step %d' % (r, t)
                checked += 1
    tg.finish()
    dist.barrier()
    if rank == 0:
        assert checked == T * world
        print('DIST_CHECK_OK steps=%d world=%d' % (T, world), flush=True)
    dist.destroy_process_group()


def span_main(cfg, eng, full, tape, per, T, n_total, lo, hi, rank, world, device):
    """The same check for rollouts that run as spans (bench.py's default): runs of steps that end with an episode, their records
    back to back in one buffer, ONE gather per run, the learner rebuilding every step of a run afterwards."""
    ep = cfg.episode_length
    sg = SpanGather(ep, per, cfg.N, cfg.obs_dim, device, dst=0, depth=2, episode_words=eng.episode_record_words,
                    graph_words=eng.step_record_words)
    eng.reset()
    eng.pack_episode(out=sg.episode_record())
    sg.submit_episode()
    my_tape = tape[:, lo:hi].contiguous()
    t, c, checked = 0, 0, 0
    ref = []
    while t < T:
        k = min(ep - t % ep, T - t)
        rec = sg.span_record(c)
        eng.use_outputs(sg.output_set(eng, c))
        eng.step_span(my_tape[t:t + k], strides=rec.strides)
        sg.submit_span(c, k)
        ended = eng.episode_started
        if rank == 0:
            ref = []
            for j in range(k):
                f_obs, _, f_node, f_adj, f_rew, f_done, _ = full.step(tape[t + j], auto_reset=True)
                ref.append((f_obs.clone(), f_node.clone(), f_adj[:, 0].clone(), f_rew.clone(), f_done.clone()))
            assert full.episode_started == ended
        old_episode = sg.gathered_episode() if rank == 0 else None     # the record the run's steps (but the last, if it ended) belong to
        if ended:
            eng.pack_episode(out=sg.episode_record())
            sg.submit_episode()
        sg.span_pending[c % 2].wait() if sg.span_pending[c % 2] is not None else None
        if rank == 0:
            torch.cuda.synchronize(device)
            new_episode = sg.gathered_episode()
            for r, (obs, rew, done, graph) in enumerate(sg.gathered_span(c)):
                l2, h2 = shard_range(n_total, world, r)
                for j in range(k):
                    f_obs, f_node, f_adj, f_rew, f_done = ref[j]
                    assert torch.equal(obs[j], f_obs[l2:h2]), 'obs of rank %d at step %d' % (r, t + j)
                    assert torch.equal(rew[j], f_rew[l2:h2]) and torch.equal(done[j].bool(), f_done[l2:h2].bool())
                    epi = new_episode if (ended and j == k - 1) else old_episode
                    node, adj = eng.rebuild_graph(obs[j].contiguous(), epi[r], step_record=graph[j].contiguous() if graph is not None else None)
                    assert torch.equal(node, f_node[l2:h2]), 'rebuilt node_obs of rank %d at step %d' % (r, t + j)
                    assert torch.equal(adj, f_adj[l2:h2]), 'rebuilt adj of rank %d at step %d' % (r, t + j)
                    checked += 1
        t += k
        c += 1
    sg.finish()
    dist.barrier()
    if rank == 0:
        assert checked == T * world
        print('DIST_CHECK_OK steps=%d world=%d' % (T, world), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
