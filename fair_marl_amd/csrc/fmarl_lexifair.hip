// Lexicographic-bottleneck ("lexifair") goal assignment on gfx950, one lane group per environment.
//
// Restates solve_fair_assignment (reference marl_fair_assign.py:16-55: N rounds of
// "minimise the largest assigned cost, fix the row that attains it"), which the reference solves
// with pyomo + Gurobi MILPs.  For distinct costs the result is the unique assignment whose
// descending-sorted cost vector is lexicographically minimal; ties are broken here by the total
// order (cost, row * G + col).
//
// Algorithm (all state in registers, cross-lane traffic by DPP / ds_bpermute shuffles only):
//   lane r of a G-lane group owns row r of the N x N cost matrix (N <= G, G in {4, 8, 16, 32, 64}).
//   Keep a perfect matching of the still-free rows/cols (start: rows greedily take their cheapest free col).  Repeat: take the
//   matched edge (r*, c*) with the largest key; drop it and look for an augmenting path from r* to
//   c* through edges with a strictly smaller key (alternating BFS on 64-bit row masks).  Found ->
//   the matching's largest key strictly decreased.  Not found -> by Berge's theorem no perfect
//   matching avoids (r*, c*) below that key, so it is the bottleneck edge: fix it, retire r*, c*.
#pragma once
#include "fmarl_dev.h"
#include "fmarl_kernels.h"

namespace fmarl {

template <int G> struct MaskOf { using type = uint64_t; };
template <> struct MaskOf<4> { using type = uint32_t; };
template <> struct MaskOf<8> { using type = uint32_t; };
template <> struct MaskOf<16> { using type = uint32_t; };
template <> struct MaskOf<32> { using type = uint32_t; };

// row / column sets: OR over the group, by DPP inside 16-lane rows + one shuffle per doubling beyond (fmarl_dev.h)
template <int G> __device__ __forceinline__ uint32_t group_or(uint32_t v) { return group_or32<G>(v); }
template <int G> __device__ __forceinline__ uint64_t group_or(uint64_t v) {
    return ((uint64_t)group_or32<G>((uint32_t)(v >> 32)) << 32) | group_or32<G>((uint32_t)v);
}
__device__ __forceinline__ int lowest_bit(uint32_t m) { return __builtin_ctz(m); }
__device__ __forceinline__ int lowest_bit(uint64_t m) { return __builtin_ctzll(m); }

// The solve itself: lane `lane` of a G-lane group holds row `lane` of the cost matrix in c[0..G) (entries
// beyond N are ignored).  Returns the column assigned to row `lane` (valid for lane < N).
// `warm` (round 6): a column for row `lane` to START from -- the previous step's assignment, where costs moved by one step (reward(agent 0)
// re-assigns every step, nav_fairassign...py:704-721).  Any perfect matching is a valid start and the result does not depend on it (the
// optimum under the total order is unique); the previous optimum is still optimal in 60 % of the steps at 10 agents and one augmentation
// away otherwise: 16.5 -> 11.0 outer iterations, 28.6 -> 17.3 BFS levels per solve, and the greedy start's N rounds are gone
// (tools/lexifair_sim.py).  Used only if the lanes' values form a permutation of the N columns; -1 (or anything else): the greedy start.
template <int G>
__device__ int lexifair_group(const double (&c)[G], int N, int warm = -1) {
    using M = typename MaskOf<G>::type;   // row / column sets of one group
    const M one = 1;
    const int lane = threadIdx.x % G;
    const M full = N >= (int)(8 * sizeof(M)) ? ~(M)0 : (M)((one << N) - 1);

    // initial perfect matching: rows in order take their cheapest free column (any perfect matching is a
    // valid start; this one needs about a third fewer improvement rounds than the identity)
    int mc = lane, mr = lane;       // col matched to row `lane`; row matched to col `lane`
    double mycost = 0.0;
    const bool in_cols = lane < N && warm >= 0 && warm < N;
    if (group_or<G>(in_cols ? (M)(one << warm) : (M)0) == full) {   // a permutation of the columns: start from it
        mc = lane < N ? warm : lane;
        for (int r = 0; r < N; ++r) {
            const int pick = __shfl(mc, r, G);
            if (lane == pick) mr = r;
        }
#pragma unroll
        for (int j = 0; j < G; ++j) if (j == mc) mycost = c[j];
    } else {
        M freec = full;
        for (int r = 0; r < N; ++r) {
            double best = __builtin_huge_val();
            int bj = 0;
#pragma unroll
            for (int j = 0; j < G; ++j) {   // branch-free: selects, no exec-mask games per column
                const bool take = (bool)((freec >> j) & 1) & (c[j] < best);
                best = take ? c[j] : best;
                bj = take ? j : bj;
            }
            const int pick = __shfl(bj, r, G);
            if (lane == r) { mc = bj; mycost = best; }
            if (lane == pick) mr = r;
            freec &= ~(one << pick);
        }
    }
    M R = full, C = full;    // free rows / cols (uniform over the group)

    for (int iter = 0; R != 0 && iter < N * N + N + 8; ++iter) {
        // 1. matched edge with the largest key (cost, row * G + col) among the free rows
        //    (equal costs: the highest row wins, which is the largest row * G + col)
        const bool arow = (R >> lane) & 1;
        double kc;
        int rstar;
        group_argmax<G>(mycost, arow, kc, rstar);
        const int cstar = __shfl(mc, rstar, G), kidx = rstar * G + cstar;
        // 2. edges of this row with a strictly smaller key, restricted to the free columns
        //    (branch-free: one mask of the cheaper entries, one of the equally expensive ones, of which
        //    those with a smaller column index than thr = kidx - lane * G count)
        M lt = 0, eq = 0;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            lt |= (M)(c[j] < kc) << j;
            eq |= (M)(c[j] == kc) << j;
        }
        const int thr = kidx - lane * G;
        const M below = thr <= 0 ? (M)0 : (thr >= G ? ~(M)0 : (M)((one << (thr & (G - 1))) - 1));
        const M adj = arow ? (lt | (eq & below)) & C : (M)0;
        // 3. alternating BFS from the freed row r* to the freed column c*; rows and columns remember the level
        //    they were reached at, the path itself is only traced when there is one
        M F = one << rstar, VC = 0;
        int rlvl = lane == rstar ? 0 : -1, clvl = -1;
        bool found = false;
        for (int lvl = 0; lvl < N; ++lvl) {
            const M nc = group_or<G>(((F >> lane) & 1) ? adj : (M)0) & ~VC;
            if (nc == 0) break;
            const bool newcol = (nc >> lane) & 1;
            if (newcol) clvl = lvl;
            VC |= nc;
            if ((nc >> cstar) & 1) { found = true; break; }
            F = group_or<G>(newcol ? (M)(one << mr) : (M)0);
            if ((F >> lane) & 1) rlvl = lvl + 1;
        }
        if (found) {   // flip a shortest path back from c*: every matched key is now smaller than the old maximum
            int ccur = cstar;
            for (int hop = 0; hop < N; ++hop) {
                const int lc = __shfl(clvl, ccur, G);   // ccur was reached from some row of level lc
                const M cand = group_or<G>((rlvl == lc && ((adj >> ccur) & 1)) ? (M)(one << lane) : (M)0);
                const int r = lowest_bit(cand);
                const int cprev = __shfl(mc, r, G);
                if (lane == r) mc = ccur;
                if (lane == ccur) mr = r;
                if (r == rstar) break;
                ccur = cprev;
            }
#pragma unroll
            for (int j = 0; j < G; ++j) if (j == mc) mycost = c[j];
        } else {       // (r*, c*) is the bottleneck edge of the remaining problem: fix it
            R &= ~(one << rstar);
            C &= ~(one << cstar);
        }
    }
    return mc;
}

template <int G>
__global__ __launch_bounds__(256) void lexifair_kernel(const double *costs, const double2 *agent_pos,
                                                       const double2 *goal_pos, int32_t *perm,
                                                       const int *flag, int n_envs, int N) {
    const int lane = threadIdx.x % G;
    const int grp = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (grp >= n_envs || (flag != nullptr && flag[grp] == 0)) return;   // group-uniform: whole groups leave
    const int env = grp;
    const bool is_row = lane < N;
    double c[G];   // row `lane` of the cost matrix (navigation_graph.py:555 cdist when built from positions)
#pragma unroll
    for (int j = 0; j < G; ++j) {
        double v = 0.0;
        if (is_row && j < N)
            v = costs ? costs[((size_t)env * N + lane) * N + j]
                      : dist2(agent_pos[(size_t)env * N + lane], goal_pos[(size_t)env * N + j]);
        c[j] = v;
    }
    const int mc = lexifair_group<G>(c, N);
    if (is_row) perm[(size_t)env * N + lane] = mc;
}

template <int G>
static void launch_lexifair_g(const double *costs, const double2 *ap, const double2 *gp, int32_t *perm,
                              const int *flag, int n_envs, int N, hipStream_t stream) {
    const int per_block = 256 / G;
    const int blocks = (n_envs + per_block - 1) / per_block;
    hipLaunchKernelGGL(lexifair_kernel<G>, dim3(blocks), dim3(256), 0, stream, costs, ap, gp, perm, flag, n_envs, N);
}

static void launch_lexifair_any(const double *costs, const double2 *ap, const double2 *gp, int32_t *perm,
                                const int *flag, int n_envs, int N, hipStream_t stream) {
    if (N <= 4) launch_lexifair_g<4>(costs, ap, gp, perm, flag, n_envs, N, stream);
    else if (N <= 8) launch_lexifair_g<8>(costs, ap, gp, perm, flag, n_envs, N, stream);
    else if (N <= 16) launch_lexifair_g<16>(costs, ap, gp, perm, flag, n_envs, N, stream);
    else if (N <= 32) launch_lexifair_g<32>(costs, ap, gp, perm, flag, n_envs, N, stream);
    else launch_lexifair_g<64>(costs, ap, gp, perm, flag, n_envs, N, stream);
}

void launch_lexifair_costs(const double *costs, int32_t *perm, int n_envs, int N, hipStream_t stream) {
    launch_lexifair_any(costs, nullptr, nullptr, perm, nullptr, n_envs, N, stream);
}

// reset path: costs = cdist(agent_pos, landmark_pos) of the envs flagged by reset_place_kernel
void launch_lexifair_state(const Params &p, hipStream_t stream) {
    launch_lexifair_any(nullptr, p.agent_pos, p.landmark_pos, p.goal_match, p.reset_flag, p.n_envs, p.N, stream);
}

}  // namespace fmarl
