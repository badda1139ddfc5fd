"""measurement aid (CPU): how unevenly the NEAR contact pairs (the float64 softplus evaluations of agent_force, fmarl_step.hip) fall on the
lanes of a wave -- oracle trajectories of navigation_graph with random actions; per wave and step the rounds agent_force takes (the busiest
lane's near partners, per block of 32 partners) against ceil(pairs / 64) if the pairs were spread over the lanes.
usage: python tools/near_pairs_sim.py"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nav_oracle as no
from oracle.philox import PhiloxStream
def run(N, L, O, n, T=20):
    cfg = no.Config(num_agents=N, num_landmarks=L, num_obstacles=O)
    env = no.OracleGraphVecEnv(cfg, n, mode='subproc', streams=lambda e, ep: PhiloxStream(3, e, ep))
    env.reset()
    rs = np.random.RandomState(0)
    epb = 256 // N
    res = []
    for t in range(T):
        env.step(rs.randint(0, 5, size=(n, N)))
        if t < 3: continue
        st = env.st
        ap = st.agent_pos; ob = st.obstacle_pos
        P = np.concatenate([ap, ob], 1)           # partners (no walls in these configs)
        d = np.sqrt(((ap[:, :, None, :] - P[:, None, :, :]) ** 2).sum(-1))
        near = (d <= 0.1 + 24 * 0.02)
        for i in range(N): near[:, i, i] = False
        cnt = near.sum(-1)    # (n, N)
        # blocks of 32 partners
        nb = (P.shape[1] + 31) // 32
        for w0 in range(0, n - n % epb, epb):
            lanes = cnt[w0:w0 + epb].reshape(-1)
            lanes = np.concatenate([lanes, np.zeros(256 - len(lanes), int)])
            nearb = near[w0:w0 + epb].reshape(-1, P.shape[1])
            nearb = np.concatenate([nearb, np.zeros((256 - len(nearb), P.shape[1]), bool)])
            for w in range(4):
                old = new = 0
                for b in range(nb):
                    c = nearb[64 * w:64 * w + 64, 32 * b:32 * b + 32].sum(-1)
                    old += c.max(); new += -(-c.sum() // 64)
                res.append((old, new, lanes[64 * w:64 * w + 64].mean()))
    r = np.array(res, float)
    print('N=%d: near partners per agent %.2f; per wave and step: rounds of the f64 pair evaluation today (max over lanes, per block) %.2f, balanced %.2f' % (N, r[:, 2].mean(), r[:, 0].mean(), r[:, 1].mean()))
run(10, 10, 3, 100)
run(32, 32, 8, 32)
run(3, 3, 3, 85)
