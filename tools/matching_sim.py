"""measurement aid (CPU, build container or GPU box): how much work the warm-started slot matchings of fair_graph_formation inherently
are.  Trajectories of the formation oracle (10 agents, random actions), then the device's algorithm (fmarl_formation.hip
hungarian_pair: potentials of the previous matching, every row claims the column of its reduced minimum, the rows left over go through
the augmenting search) replayed in NumPy -- it reproduces the device's own counters (FMARL_MEASURE build, tools/phase_ticks.py cfg4:
2.49 rows / 9.38 search iterations per matching on the current slots, 2.30 / 8.05 on the previous ones) -- next to (a) how often the
OPTIMAL assignment itself changes from one step to the next and (b) what centred potentials (one or three sweeps that balance every
matched pair's row slack against its column slack) would buy.

Round 5: also at WAVE level -- six envs share a wave, and loops that the envs walk in lockstep run as long as the env that needs them
longest: the nested form of round 4 (per unmatched row: search iterations, then path columns) against the flattened form (every
env its own state machine: the wave runs as long as its busiest env's total).

usage: python tools/matching_sim.py [envs=48] [steps=40]"""
import os
import sys

import numpy as np
from scipy.optimize import linear_sum_assignment

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import formation_oracle as fo  # noqa: E402
from oracle.philox import PhiloxStream  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N = 10
cfg = fo.Config(num_agents=N, num_landmarks=1, num_obstacles=3)
env = fo.OracleFormationVecEnv(cfg, n, mode='subproc', streams=lambda e, ep: PhiloxStream(3, e, ep))
env.reset()
rs = np.random.RandomState(0)
X, S = [env.st.agent_pos.copy()], [env.st.slot_pos.copy()]
for t in range(T):
    env.step(rs.randint(0, 5, size=(n, N)))
    X.append(env.st.agent_pos.copy())
    S.append(env.st.slot_pos.copy())
X, S = np.array(X), np.array(S)   # (T + 1, n, N, 2)
print('mean |motion| per step: agents %.3f, slots %.3f (slot spacing on the ring 0.31)' % (np.abs(X[1:] - X[:-1]).mean(), np.abs(S[1:] - S[:-1]).mean()))


def cost(x, P):
    return np.sqrt(((x[:, None, :] - P[None, :, :]) ** 2).sum(-1))


ROOTS = []   # per call: [(search iterations, path columns) per unmatched row] -- the wave-level figures below


def hungarian(c, v):
    """The device's matching: -> (ans[row] = col, u, v, rows through the augmenting search, search iterations)."""
    v = v.copy()
    red = c - v[None, :]
    u, mine = red.min(1), red.argmin(1)
    prow = -np.ones(N, int)
    for r in range(N):
        if prow[mine[r]] < 0:
            prow[mine[r]] = r
    um = [r for r in range(N) if prow[mine[r]] != r]
    nit = 0
    roots = []
    for i in um:
        nit0 = nit
        minv, way, used, intree = np.full(N, 1e300), -np.ones(N, int), np.zeros(N, bool), np.zeros(N, bool)
        intree[i] = True
        i0, j0 = i, -1
        while True:
            nit += 1
            cur = c[i0] - u[i0] - v
            upd = (~used) & (cur < minv)
            minv[upd], way[upd] = cur[upd], j0
            mm = np.where(~used, minv, 1e300)
            j1 = int(mm.argmin())
            delta = mm[j1]
            u[intree] += delta
            v[used] -= delta
            minv[~used] -= delta
            used[j1] = True
            j0 = j1
            r1 = prow[j1]
            if r1 < 0:
                break
            intree[r1] = True
            i0 = r1
        j, nflip = j1, 0
        while j >= 0:
            jp = way[j]
            prow[j] = i if jp < 0 else prow[jp]
            j = jp
            nflip += 1
        roots.append((nit - nit0, nflip))
    ROOTS.append(roots)
    ans = np.empty(N, int)
    ans[prow] = np.arange(N)
    return ans, u, v, len(um), nit


def centre(c, ans, u, v, sweeps):
    """Every matched pair (i, ans[i]) moves delta = (R_i - S_j) / 2 from its column potential to its row potential: R_i = the row's
    least slack to another column, S_j = the column's least slack from another row.  Feasible, the matching stays tight."""
    u, v = u.copy(), v.copy()
    inv = np.empty(N, int)
    inv[ans] = np.arange(N)
    for _ in range(sweeps):
        sl = c - u[:, None] - v[None, :]
        R = np.array([np.min(np.delete(sl[i], ans[i])) for i in range(N)])
        Sj = np.array([np.min(np.delete(sl[:, j], inv[j])) for j in range(N)])
        dl = (R - Sj[ans]) / 2
        u, v[ans] = u + dl, v[ans] - dl
    assert (c - u[:, None] - v[None, :]).min() > -1e-12
    return v


changed = [0, 0, 0, 0, 0]
for e in range(n):
    prev = None
    for t in range(1, T + 1):
        a0 = linear_sum_assignment(cost(X[t, e], S[t, e]))[1]
        a1 = linear_sum_assignment(cost(X[t, e], S[t - 1, e]))[1]
        if prev is not None and t % 25 not in (0, 1, 2):
            changed[0] += (a0 != prev).sum(); changed[1] += (a1 != prev).sum(); changed[2] += (a0 == prev).all(); changed[3] += (a1 == prev).all(); changed[4] += 1
        prev = a0
print('the OPTIMAL assignment from one step to the next: %.2f rows change their column (current slots), %.2f (previous slots); unchanged in '
      '%.0f %% / %.0f %% of the steps' % (changed[0] / changed[4], changed[1] / changed[4], 100 * changed[2] / changed[4], 100 * changed[3] / changed[4]))
for sweeps in (0, 1, 3):
    tot = np.zeros(5)
    for e in range(n):
        v = np.zeros(N)
        for t in range(1, T + 1):
            if t % 25 == 1 and t > 1:
                v = np.zeros(N)   # a fresh episode
            c0, c1 = cost(X[t, e], S[t, e]), cost(X[t, e], S[t - 1, e])
            _, _, _, na1, ni1 = hungarian(c1, v)
            a0, u0, v0, na0, ni0 = hungarian(c0, v)
            assert (linear_sum_assignment(c0)[1] == a0).all()
            if sweeps:
                v0 = centre(c0, a0, u0, v0, sweeps)
            v = v0 - v0.max()
            if t % 25 not in (0, 1, 2):
                tot += (na0, ni0, na1, ni1, 1)
    print('%-28s current slots: %.2f rows through the search, %.2f iterations;  previous slots: %.2f rows, %.2f iterations'
          % ('potentials as left' if not sweeps else 'centred, %d sweep(s)' % sweeps, tot[0] / tot[4], tot[1] / tot[4], tot[2] / tot[4], tot[3] / tot[4]))

# ---- wave level: six consecutive envs share a wave
ROOTS.clear()
per = {}
for e in range(n - n % 6):
    v = np.zeros(N)
    for t in range(1, T + 1):
        if t % 25 == 1 and t > 1:
            v = np.zeros(N)
        c0, c1 = cost(X[t, e], S[t, e]), cost(X[t, e], S[t - 1, e])
        k0 = len(ROOTS)
        hungarian(c1, v)
        a0, u0, v0, _, _ = hungarian(c0, v)
        v = v0 - v0.max()
        per[(t, e)] = (ROOTS[k0 + 1], ROOTS[k0])   # (current slots, previous slots)
nested_it = nested_fl = flat = cnt = mean_it = mean_fl = 0
for t in range(3, min(T, 24) + 1):
    for w in range((n - n % 6) // 6):
        envs = range(6 * w, 6 * w + 6)
        for which in (0, 1):
            rr = [per[(t, e)][which] for e in envs]
            for k in range(max(len(r) for r in rr)):
                nested_it += max((r[k][0] if k < len(r) else 0) for r in rr)
                nested_fl += max((r[k][1] if k < len(r) else 0) for r in rr)
        tot = [sum(a for a, b in per[(t, e)][0]) + sum(a for a, b in per[(t, e)][1]) for e in envs]
        flat += max(tot)
        mean_it += np.mean(tot)
        mean_fl += np.mean([sum(b for a, b in per[(t, e)][0]) + sum(b for a, b in per[(t, e)][1]) for e in envs])
        cnt += 1
print('per wave and step (both matchings): nested loops in lockstep %.1f search trips + %.1f path trips; flattened (path flipped in one step) '
      '%.1f trips; one env needs %.1f search iterations + %.1f path columns' % (nested_it / cnt, nested_fl / cnt, flat / cnt, mean_it / cnt, mean_fl / cnt))
