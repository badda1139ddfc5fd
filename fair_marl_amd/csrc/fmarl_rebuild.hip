// Learner-side reconstruction of node_obs / adj from the gathered trajectory record (navigation_graph).
//
// The reference's workers pipe every step's (obs, agent_id, node_obs, adj, ...) to the learner
// (reference onpolicy/envs/env_wrappers.py:988-996).  Across GPUs only the compact part travels: the
// per-step obs rows (velocity, position: navigation_graph.py:826-857) plus, once per episode, the
// entities that stay put for the whole episode -- each agent's goal, landmarks, obstacles, walls
// (navigation_graph.py:264-575: placed in reset_world, never moved by World.step).  From the two the
// learner rank rebuilds what graph_observation / _get_entity_feat_relative would have produced
// (navigation_graph.py:941-1035, 1079-1124): the same f32 tables go into LDS, the same emission runs.
// node_obs is bit-identical to the sender's (the sender's emission starts from the same f32 roundings);
// adj likewise: the sender computes it from the same f32 position table.
//
// Episode record, 32-bit words per env:  goal (gx, gy) f32 x N | landmark, obstacle (x, y) f32 x (L + O) |
// wall [axis f64 (2 words), e0 f32, e1 f32, orient f32, 0] x W.
#pragma once
#include "fmarl_dev.h"
#include "fmarl_kernels.h"

namespace fmarl {

__host__ __device__ inline int episode_record_words(int N, int L, int O, int W) { return 2 * N + 2 * (L + O) + 6 * W; }

__global__ __launch_bounds__(256) void pack_episode_kernel(Params p, uint32_t *rec) {
    const int LO = p.L + p.O, slots = p.N + LO + p.W, words = episode_record_words(p.N, p.L, p.O, p.W);
    const size_t total = (size_t)p.n_envs * slots;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int env = (int)(t / slots), k = (int)(t - (size_t)env * slots);
        uint32_t *r = rec + (size_t)env * words;
        if (k < p.N + LO) {
            double2 x;
            if (k < p.N) x = p.scenario == FMARL_SCENARIO_NAVIGATION_GRAPH ? p.landmark_pos[(size_t)env * p.L + p.goal_match[(size_t)env * p.N + k]]
                                                                         : make_double2(0.0, 0.0);   // (goals travel per step elsewhere)
            else if (k < p.N + p.L) x = p.landmark_pos[(size_t)env * p.L + (k - p.N)];
            else x = p.obstacle_pos[(size_t)env * p.O + (k - p.N - p.L)];
            ((float *)r)[2 * k] = (float)x.x;
            ((float *)r)[2 * k + 1] = (float)x.y;
        } else {
            const int w = k - p.N - LO;
            const size_t g = (size_t)env * p.W + w;
            uint32_t *q = r + 2 * (p.N + LO) + 6 * w;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(p.wall_axis[g]);
            q[0] = (uint32_t)bits; q[1] = (uint32_t)(bits >> 32);
            ((float *)q)[2] = (float)p.wall_e0[g]; ((float *)q)[3] = (float)p.wall_e1[g];
            ((float *)q)[4] = (float)p.wall_orient[g]; q[5] = 0;
        }
    }
}

// Same workgroup shape and LDS tables as step_kernel / reset_emit_kernel; n_envs is the number of envs
// in (obs, rec), not the handle's.
__global__ __launch_bounds__(kThreads) void rebuild_graph_kernel(Params p, FmarlOutputs o, const float *obs,
                                                                 const uint32_t *rec, int n_envs) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int env0 = env_block(p) * p.epb;
    const int nenv = min(p.epb, n_envs - env0);
    const int LO = p.L + p.O, words = episode_record_words(p.N, p.L, p.O, p.W);
    const int el = tid / p.N, i = tid - el * p.N;
    if (el < nenv) {
        char *base = lds + (size_t)el * p.lds_env_bytes;
        const size_t g = (size_t)(env0 + el) * p.N + i;
        const float *ob = obs + g * p.D;   // [vx vy x y ...]: navigation_graph.py:855-857
        const float *goal = (const float *)(rec + (size_t)(env0 + el) * words) + 2 * i;
        const double2 x = make_double2((double)ob[2], (double)ob[3]);
        ((double2 *)(base + p.lds_pos))[i] = x;
        store_agent_rows(p, base, i, x, make_double2((double)ob[0], (double)ob[1]), make_double2((double)goal[0], (double)goal[1]));
        if (i == 0) *(int *)(base + p.lds_flag) = 0;
    }
    for (int t = tid; t < nenv * LO; t += kThreads) {
        const int e_l = t / LO, k = t - e_l * LO;
        const float *s = (const float *)(rec + (size_t)(env0 + e_l) * words) + 2 * (p.N + k);
        ((double2 *)(lds + (size_t)e_l * p.lds_env_bytes + p.lds_pos))[p.N + k] = make_double2((double)s[0], (double)s[1]);
        ((float2 *)(lds + (size_t)e_l * p.lds_env_bytes + p.lds_posf))[p.N + k] = make_float2(s[0], s[1]);
    }
    for (int t = tid; t < nenv * p.W; t += kThreads) {
        const int e_l = t / p.W, w = t - e_l * p.W;
        char *base = lds + (size_t)e_l * p.lds_env_bytes;
        const uint32_t *q = rec + (size_t)(env0 + e_l) * words + 2 * (p.N + LO) + 6 * w;
        const double axis = __longlong_as_double((long long)((unsigned long long)q[0] | ((unsigned long long)q[1] << 32)));
        const float *qf = (const float *)q;
        double *wl = (double *)(base + p.lds_wall) + w * 4;
        wl[0] = axis; wl[1] = (double)qf[2]; wl[2] = (double)qf[3]; wl[3] = (double)qf[4];
        ((double2 *)(base + p.lds_pos))[p.N + LO + w] = qf[4] == 0.f ? make_double2(0.0, axis) : make_double2(axis, 0.0);
        ((float2 *)(base + p.lds_posf))[p.N + LO + w] = qf[4] == 0.f ? make_float2(0.f, (float)axis) : make_float2((float)axis, 0.f);
        ((float4 *)(base + p.lds_wallf))[w] = make_float4((float)wl[1], (float)(axis + kWallWidth / 2), (float)wl[2], (float)(axis - kWallWidth / 2));
    }
    for (int t = tid; t < nenv; t += kThreads)
        *(float4 *)(lds + (size_t)t * p.lds_env_bytes + p.lds_constf) = make_float4(0.f, 1.f, 2.f, 3.f);
    __syncthreads();
    emit_graph(p, o, lds, env0, nenv);
}

}  // namespace fmarl
