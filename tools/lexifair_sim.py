"""measurement aid (CPU): the device's lexifair algorithm (fmarl_lexifair.hip lexifair_group: a perfect matching improved by augmenting
paths below the current bottleneck key until every row is fixed) replayed on oracle trajectories of nav_fairassign_fairrew_formation_graph --
outer iterations and BFS levels per solve when the start is the greedy matching (what the kernel does) and when it is the PREVIOUS
step's assignment (reward(agent 0) re-assigns every step on positions that moved by one step).
usage: python tools/lexifair_sim.py [N=10] [envs=16] [steps=60]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fairnav_oracle as fnv  # noqa: E402
from oracle.lexifair import lexifair  # noqa: E402
from oracle.philox import PhiloxStream  # noqa: E402


def solve(c, start=None):
    """-> (assignment, outer iterations, BFS levels walked, augmentations)"""
    N = c.shape[0]
    key = lambda r, j: (c[r, j], r * 64 + j)  # noqa: E731
    if start is None:   # rows in order take their cheapest free column
        mc, free = [-1] * N, set(range(N))
        for r in range(N):
            j = min(free, key=lambda j: (c[r, j], j))
            mc[r] = j; free.discard(j)
    else:
        mc = list(start)
    mr = [0] * N
    for r in range(N):
        mr[mc[r]] = r
    R, C = set(range(N)), set(range(N))
    iters = levels = augs = 0
    while R:
        iters += 1
        rstar = max(R, key=lambda r: key(r, mc[r]))
        cstar = mc[rstar]
        k = key(rstar, cstar)
        adj = {r: {j for j in C if key(r, j) < k} for r in R}
        F, VC, parent_c, found = {rstar}, set(), {}, False
        rl = {rstar: None}
        while F:
            levels += 1
            nc = set()
            for r in F:
                for j in adj[r]:
                    if j not in VC and j not in nc:
                        nc.add(j); parent_c[j] = r
            if not nc:
                break
            VC |= nc
            if cstar in nc:
                found = True
                break
            F = set()
            for j in nc:
                r = mr[j]
                if r not in rl:
                    rl[r] = j; F.add(r)
        if found:
            augs += 1
            ccur = cstar
            while True:
                r = parent_c[ccur]
                cprev = mc[r]
                mc[r] = ccur; mr[ccur] = r
                if r == rstar:
                    break
                ccur = cprev
        else:
            R.discard(rstar); C.discard(cstar)
    return mc, iters, levels, augs


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    cfg = fnv.Config(scenario_name='nav_fairassign_fairrew_formation_graph', num_agents=N, num_landmarks=N, num_obstacles=3, goal_rew=30.0, collision_rew=30.0)
    env = fnv.OracleFairNavVecEnv(cfg, n, mode='subproc', streams=lambda e, ep: PhiloxStream(1, e, ep))
    env.reset()
    rs = np.random.RandomState(0)
    tot = {'greedy': np.zeros(3), 'warm': np.zeros(3)}
    solves = changed = 0
    prev = [None] * n
    for t in range(steps):
        env.step(rs.randint(0, 5, size=(n, N)))
        for e in range(n):
            a, g = env.st.agent_pos[e], env.st.landmark_pos[e]
            c = np.sqrt(((a[:, None, :] - g[None, :, :]) ** 2).sum(-1))
            m0, i0, l0, a0 = solve(c)
            assert list(m0) == list(lexifair(c)), 'the replay must equal the oracle'
            tot['greedy'] += (i0, l0, a0)
            if prev[e] is not None:
                m1, i1, l1, a1 = solve(c, prev[e])
                assert m1 == m0
                tot['warm'] += (i1, l1, a1)
                changed += m0 != list(prev[e])
                solves += 1
            prev[e] = m0
    print('N = %d, %d solves; the assignment changed from one step to the next in %.1f %% of them' % (N, solves, 100.0 * changed / max(1, solves)))
    for k in ('greedy', 'warm'):
        d = tot[k] / (solves if k == 'warm' else n * steps)
        print('%-6s start: %.2f outer iterations, %.2f BFS levels, %.2f augmentations per solve' % (k, d[0], d[1], d[2]))


if __name__ == '__main__':
    main()
