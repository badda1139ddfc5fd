// Fused rollout step for gfx950: one thread per (env, agent), kThreads per workgroup.
//
// Restates, per environment (reference paths relative to the reference repo root):
//   multiagent/environment.py:816-877   MultiAgentGraphEnv.step ordering
//   multiagent/environment.py:265-311   _set_action (u = 5 * [a1 - a2, a3 - a4])
//   multiagent/core.py:250-462          World.step: action force, entity + wall collision forces, integrate
//   multiagent/core.py:204-228          calculate_distances (-> adj)
//   multiagent/custom_scenarios/navigation_graph.py:826-857, 760-824, 941-1035, 1079-1124, 577-647
//                                       observation, reward, graph_observation, info_callback
//
// Data flow inside one workgroup (epb = 256 / N environments):
//   HBM state (f64 SoA, 16 B per lane for pos / vel) -> registers + LDS entity table
//   -> all-pairs soft-contact forces out of LDS -> integrate in registers
//   -> LDS: new positions, per-env fairness vectors -> per-agent obs / reward / info (closed form of the
//      reference's sequential agent loop) -> LDS row tables A (entity features) and B (ego features)
//   -> streaming, lane-contiguous stores of node_obs = A - B and adj = |x_a - x_b| to HBM.
#pragma once
#include "fmarl_dev.h"
#include "fmarl_kernels.h"

namespace fmarl {

// ------------------------------------------------------------------------------------------------
// Emission: node_obs[env][i][e][f] = A[env][e][f] - B[env][i][bsel[f]];  adj[env][a][b] = |x_a - x_b|.
// Lanes walk the flat output index so that every wave store is 256 contiguous bytes.
// `skip` (per env, in LDS) marks envs whose observation is produced by the reset path instead.
__device__ void emit_graph(const Params &p, const FmarlOutputs &o, char *lds, int env0, int nenv) {
    const int tid = threadIdx.x;
    const uint32_t NEF = p.N * p.E * p.F, EF = p.E * p.F, EE = p.E * p.E;
    if (o.node_obs) {
        float *dst = o.node_obs + (size_t)env0 * NEF;
        const uint32_t total = nenv * NEF;
        for (uint32_t q = tid; q < total; q += kThreads) {
            uint32_t el = p.dNEF.div(q);
            const char *base = lds + (size_t)el * p.lds_env_bytes;
            if (*(const int *)(base + p.lds_flag)) continue;
            uint32_t r = q - el * NEF;
            uint32_t i = p.dEF.div(r);
            uint32_t s = r - i * EF;
            uint32_t f = s - p.dF.div(s) * p.F;
            const float *A = (const float *)(base + p.lds_a);
            const float *B = (const float *)(base + p.lds_b);
            dst[q] = A[s] - B[i * kBWidth + (uint32_t)((p.bsel >> (4 * f)) & 0xF)];
        }
    }
    if (o.adj) {
        float *dst = o.adj + (size_t)env0 * EE;
        const uint32_t total = nenv * EE;
        for (uint32_t q = tid; q < total; q += kThreads) {
            uint32_t el = p.dEE.div(q);
            const char *base = lds + (size_t)el * p.lds_env_bytes;
            if (*(const int *)(base + p.lds_flag)) continue;
            uint32_t r = q - el * EE;
            uint32_t a = p.dE.div(r);
            uint32_t b = r - a * p.E;
            const double2 *pos = (const double2 *)(base + p.lds_pos);
            double2 pa = pos[a], pb = pos[b];
            float dx = (float)(pa.x - pb.x), dy = (float)(pa.y - pb.y);
            dst[q] = sqrtf(dx * dx + dy * dy);
        }
    }
}

// Static rows of the A table (landmarks, obstacles, walls) and the wall records; cooperative.
// navigation_graph.py:1100-1124: [rel_vel, rel_pos, rel_goal = rel_pos, rel_pos, rel_pos, type];
// walls replace the last two pairs by corner points (:1115-1116).
__device__ void fill_static_rows(const Params &p, char *lds, int nenv) {
    const int S = p.L + p.O + p.W;
    for (int t = threadIdx.x; t < nenv * S; t += kThreads) {
        int el = t / S, k = t - el * S;
        char *base = lds + (size_t)el * p.lds_env_bytes;
        const double2 *pos = (const double2 *)(base + p.lds_pos);
        float *A = (float *)(base + p.lds_a) + (p.N + k) * p.F;
        double2 x = pos[p.N + k];
        float fx = (float)x.x, fy = (float)x.y;
        float type = k < p.L ? 1.f : (k < p.L + p.O ? 2.f : 3.f);
        A[0] = 0.f; A[1] = 0.f; A[2] = fx; A[3] = fy; A[4] = fx; A[5] = fy;
        if (k < p.L + p.O) {
            A[6] = fx; A[7] = fy; A[8] = fx; A[9] = fy;
        } else {
            const double *wl = (const double *)(base + p.lds_wall) + (k - p.L - p.O) * 4;
            A[6] = (float)wl[1]; A[7] = (float)(wl[0] + kWallWidth / 2);   // (e0, axis + w/2)
            A[8] = (float)wl[2]; A[9] = (float)(wl[0] - kWallWidth / 2);   // (e1, axis - w/2)
        }
        A[10] = type;
    }
}

// Loads landmarks / obstacles / walls of the workgroup's envs into the LDS entity table.
__device__ void load_statics(const Params &p, char *lds, int env0, int nenv) {
    const int LO = p.L + p.O;
    for (int t = threadIdx.x; t < nenv * LO; t += kThreads) {
        int el = t / LO, k = t - el * LO;
        double2 *pos = (double2 *)(lds + (size_t)el * p.lds_env_bytes + p.lds_pos);
        int env = env0 + el;
        pos[p.N + k] = k < p.L ? p.landmark_pos[(size_t)env * p.L + k]
                               : p.obstacle_pos[(size_t)env * p.O + (k - p.L)];
    }
    for (int t = threadIdx.x; t < nenv * p.W; t += kThreads) {
        int el = t / p.W, w = t - el * p.W;
        char *base = lds + (size_t)el * p.lds_env_bytes;
        size_t g = (size_t)(env0 + el) * p.W + w;
        double axis = p.wall_axis[g];
        int orient = p.wall_orient[g];
        double *wl = (double *)(base + p.lds_wall) + w * 4;
        wl[0] = axis; wl[1] = p.wall_e0[g]; wl[2] = p.wall_e1[g]; wl[3] = (double)orient;
        // wall "sphere" centre: (0, axis) for 'H', (axis, 0) for 'V' (navigation_graph.py:309-324)
        ((double2 *)(base + p.lds_pos))[p.N + LO + w] = orient == 0 ? make_double2(0.0, axis) : make_double2(axis, 0.0);
    }
}

// mean and population std of v[0..n) where entry j comes from `fresh` if j < split else from `stale`
// (np.mean / np.std, two-pass).  split = n -> all fresh.
__device__ __forceinline__ void mixed_stats(const double *fresh, const double *stale, int n, int split,
                                            double &mean, double &sd) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += (j < split) ? fresh[j] : stale[j];
    mean = s / n;
    double q = 0.0;
    for (int j = 0; j < n; ++j) {
        double d = ((j < split) ? fresh[j] : stale[j]) - mean;
        q += d * d;
    }
    sd = sqrt(q / n);
}

__global__ __launch_bounds__(kThreads) void step_kernel(Params p, FmarlOutputs o, const int32_t *action_idx,
                                                        const float *action_vec, int auto_reset) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int env0 = blockIdx.x * p.epb;
    const int nenv = min(p.epb, p.n_envs - env0);
    const int el = tid / p.N, i = tid - el * p.N;
    const bool active = el < nenv;
    const int env = env0 + el;
    const size_t g = (size_t)env * p.N + i;
    char *base = lds + (size_t)el * p.lds_env_bytes;
    double2 *s_pos = (double2 *)(base + p.lds_pos);
    double *s_stat = (double *)(base + p.lds_stat);   // [pd_new | Dg_old | Dg_new | Tr_old | Tr_new] x N

    double2 x = make_double2(0, 0), v = make_double2(0, 0);
    double pd = 0;
    int step = 0;
    if (active) {
        x = p.agent_pos[g]; v = p.agent_vel[g]; pd = p.p_dist[g];
        s_pos[i] = x;
        step = p.cur_step[env] + 1;   // environment.py:819, :823
    }
    load_statics(p, lds, env0, nenv);
    __syncthreads();

    // ---- World.step (core.py:250-274) ---------------------------------------------------------
    if (active) {
        double ux, uy;
        if (action_idx) {
            int a = action_idx[g];
            ux = kSensitivity * (double)((a == 1) - (a == 2));
            uy = kSensitivity * (double)((a == 3) - (a == 4));
        } else {
            const float *a = action_vec + g * 5;
            ux = ((double)a[1] - (double)a[2]) * kSensitivity;
            uy = ((double)a[3] - (double)a[4]) * kSensitivity;
        }
        double Fx = ux, Fy = uy;   // core.py:277-298, mass 1
        // core.py:301-316 + :370-404: agent-agent, agent-obstacle, agent-wall-entity pairs
        const int first_obst = p.N + p.L, first_wall = first_obst + p.O;
        for (int b = 0; b < p.E; ++b) {
            if (b == i || (b >= p.N && b < first_obst)) continue;   // self; landmarks do not collide
            double2 q = s_pos[b];
            double dx = x.x - q.x, dy = x.y - q.y;
            double d = sqrt(dx * dx + dy * dy);
            double dmin = b < first_wall ? 2 * kEntitySize : kEntitySize + kWallWidth;
            double z = -(d - dmin) / kContactMargin;
            if (z < -37.0) continue;   // softplus tail < 1e-16 * margin: below f64 resolution of the sum
            double pen = softplus_pen(z, kContactMargin);
            Fx += kContactForce * dx / d * pen;
            Fy += kContactForce * dy / d * pen;
        }
        // core.py:317-326 + :407-462 walls proper
        const double *wl = (const double *)(base + p.lds_wall);
        for (int w = 0; w < p.W; ++w) {
            double axis = wl[w * 4], e0 = wl[w * 4 + 1], e1 = wl[w * 4 + 2];
            bool horiz = wl[w * 4 + 3] == 0.0;
            double ppar = horiz ? x.x : x.y, pperp = horiz ? x.y : x.x;
            const double s = kEntitySize;
            if (ppar < e0 - s || ppar > e1 + s) continue;
            double theta = 0.0, dmin = s + 0.5 * kWallWidth;
            if (ppar < e0 || ppar > e1) {
                double past = ppar < e0 ? ppar - e0 : ppar - e1;
                theta = asin(past / s);
                dmin = cos(theta) * s + 0.5 * kWallWidth;
            }
            double dpos = pperp - axis, d = fabs(dpos);
            double pen = softplus_pen(-(d - dmin) / kWallContactMargin, kWallContactMargin);
            double fm = kWallContactForce * dpos / d * pen;
            double fperp = cos(theta) * fm, fpar = sin(theta) * fabs(fm);
            Fx += horiz ? fpar : fperp;
            Fy += horiz ? fperp : fpar;
        }
        // core.py:338-356 integrate
        v.x = v.x * (1 - kDamping) + Fx * kDt;
        v.y = v.y * (1 - kDamping) + Fy * kDt;
        if (p.has_max_speed) {
            double speed = sqrt(v.x * v.x + v.y * v.y);
            if (speed > p.max_speed) { v.x = v.x / speed * p.max_speed; v.y = v.y / speed * p.max_speed; }
        }
        x.x += v.x * kDt; x.y += v.y * kDt;
        double sx = v.x * kDt, sy = v.y * kDt;
        pd += sqrt(sx * sx + sy * sy);
    }
    __syncthreads();   // every lane has finished reading the old positions

    double Dg_old = 0, Tr_old = 0, left_old = 0, left_new = 0, rew = 0;
    int match = 0, noc = 0, nac = 0;
    bool will_reset = false;
    if (active) {
        s_pos[i] = x;
        Dg_old = p.dists_to_goal[g]; Tr_old = p.times_required[g]; left_old = p.dist_left[g];
        match = p.goal_match[g];
        const bool open = Tr_old == -1.0;
        s_stat[i] = pd;
        s_stat[p.N + i] = Dg_old;
        s_stat[2 * p.N + i] = open ? pd : Dg_old;                     // navigation_graph.py:590, :597
        s_stat[3 * p.N + i] = Tr_old;
        will_reset = auto_reset && step >= p.episode_length;          // env_wrappers.py:859-864
        if (i == 0) *(int *)(base + p.lds_flag) = will_reset ? 1 : 0;
    }
    __syncthreads();

    if (active) {
        const double2 goal = s_pos[p.N + match];
        const double dg = dist2(x, goal);                             // :583, :774
        const bool open = Tr_old == -1.0;
        const bool arrive = dg < p.thr && open;                       // :587
        const double Tr_new = arrive ? step * kDt : Tr_old;           // :589
        const double Dg_new = open ? pd : Dg_old;
        left_new = open ? dg : left_old;                 // :591, :598
        s_stat[4 * p.N + i] = Tr_new;

        // fairness scalar of obs_i / reward_i (:764-769, :849-854): p_dist statistics while this
        // agent's dists_to_goal is still -1, otherwise the statistics info_{i-1} left behind.
        double fairness, m, sd;
        if (Dg_old == -1.0) mixed_stats(s_stat, s_stat, p.N, p.N, m, sd);
        else mixed_stats(s_stat + 2 * p.N, s_stat + p.N, p.N, i, m, sd);
        fairness = m / (sd + 0.0001);

        // collisions (:701-705, :650-684)
        int ag_hits = 0;
        for (int j = 0; j < p.N; ++j)
            if (j != i && dist2(x, s_pos[j]) < 1.05 * (kEntitySize + kEntitySize)) ++ag_hits;
        bool ob_hit = false;
        for (int k = 0; k < p.O; ++k)
            ob_hit |= dist2(s_pos[p.N + p.L + k], x) < 1.05 * (kEntitySize + kEntitySize);
        const double *wl = (const double *)(base + p.lds_wall);
        for (int w = 0; w < p.W; ++w)
            ob_hit |= wall_box_hit(x, wl[w * 4], wl[w * 4 + 1], wl[w * 4 + 2], (int)wl[w * 4 + 3]);

        // reward (:760-824)
        rew = dg < p.thr ? p.goal_rew : -dg;
        rew -= p.collision_rew * ag_hits;
        if (ob_hit) rew -= p.collision_rew;
        double fr = p.fair_rew * tanh(fairness - p.zeroshift);
        if (fr < -2.0) fr = -2.0;
        rew += fr;
        rew = fmin(fmax(rew, -2 * p.collision_rew), p.goal_rew + p.fair_rew);

        // state + small outputs
        noc = p.num_obst_coll[g] + (ob_hit ? 1 : 0); nac = p.num_agent_coll[g] + ag_hits;
        p.agent_pos[g] = x; p.agent_vel[g] = v; p.p_dist[g] = pd;
        p.dists_to_goal[g] = Dg_new; p.times_required[g] = Tr_new; p.dist_left[g] = left_new;
        p.num_obst_coll[g] = noc; p.num_agent_coll[g] = nac;
        if (i == 0) p.cur_step[env] = step;
        if (o.reward) o.reward[g] = (float)rew;
        if (o.done) o.done[g] = step >= p.episode_length;            // environment.py:237-247
        if (o.obs && !will_reset) {                                    // :845-857
            float *ob = o.obs + g * p.D;
            ob[0] = (float)v.x; ob[1] = (float)v.y; ob[2] = (float)x.x; ob[3] = (float)x.y;
            ob[4] = (float)(goal.x - x.x); ob[5] = (float)(goal.y - x.y); ob[6] = (float)fairness;
        }
        // ego / entity rows of the graph tables (:1084-1099, :1124)
        float *A = (float *)(base + p.lds_a) + i * p.F;
        float *B = (float *)(base + p.lds_b) + i * kBWidth;
        const float fx = (float)x.x, fy = (float)x.y, fvx = (float)v.x, fvy = (float)v.y;
        A[0] = fvx; A[1] = fvy; A[2] = fx; A[3] = fy; A[4] = (float)goal.x; A[5] = (float)goal.y;
        A[6] = fx; A[7] = fy; A[8] = fx; A[9] = fy; A[10] = 0.f;
        B[0] = fvx; B[1] = fvy; B[2] = fx; B[3] = fy; B[4] = 0.f;
    }
    __syncthreads();
    if (active && o.info) {
        double dm, ds, tm, ts;
        mixed_stats(s_stat + 2 * p.N, s_stat + p.N, p.N, i + 1, dm, ds);
        mixed_stats(s_stat + 4 * p.N, s_stat + 3 * p.N, p.N, i + 1, tm, ts);
        float *inf = o.info + g * FMARL_INFO_WIDTH;
        inf[FMARL_INFO_DIST_TO_GOAL] = (float)left_new;
        inf[FMARL_INFO_TIME_REQ_TO_GOAL] = (float)s_stat[4 * p.N + i];
        inf[FMARL_INFO_NUM_AGENT_COLLISIONS] = (float)nac;
        inf[FMARL_INFO_NUM_OBST_COLLISIONS] = (float)noc;
        inf[FMARL_INFO_DISTANCE_MEAN] = (float)dm;
        inf[FMARL_INFO_DISTANCE_VARIANCE] = (float)ds;
        inf[FMARL_INFO_MEAN_BY_VARIANCE] = (float)(dm / (ds + 0.0001));
        inf[FMARL_INFO_DISTS_TRAVELED] = (float)s_stat[2 * p.N + i];
        inf[FMARL_INFO_TIME_TAKEN] = (float)(step * kDt);
        inf[FMARL_INFO_TIME_MEAN] = (float)tm;
        inf[FMARL_INFO_TIME_STDDEV] = (float)ts;
        inf[FMARL_INFO_TIME_MEAN_BY_STDDEV] = (float)(tm / (ts + 0.0001));
        inf[FMARL_INFO_MIN_TIME_TO_GOAL] = (float)p.min_time[g];
        inf[FMARL_INFO_INDIVIDUAL_REWARD] = (float)rew;
    }
    fill_static_rows(p, lds, nenv);
    __syncthreads();
    emit_graph(p, o, lds, env0, nenv);
}

}  // namespace fmarl
