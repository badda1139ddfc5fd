// nav_fairassign_fairrew_formation_graph (the scenario of the shipped FA / FA+FR
// weights) on gfx950: step + reset observation.
//
// Restates reference multiagent/custom_scenarios/nav_fairassign_fairrew_formation_graph.py ("nf:<line>")
// behind the shared World.step physics (fmarl_step.hip world_step_agent):
//   nf:691-803   reward: lexicographic-fair RE-assignment every step inside reward(agent 0) (nf:704-721),
//                stop-on-goal `status` (velocity zeroed inside reward, done per agent: environment.py:237-247,
//                no agent-agent force afterwards: core.py:394-398), fairness term floored at -fair_rew
//   nf:840-1000  observation (11): nearest / second nearest goal, float-valued goal occupancy and goal
//                history mutated in agent order
//   nf:1222-1334 graph rows (13 features); agent rows read the occupancy / history and clear the occupancy
//                when no goal is free
//   nf:489-590   info_callback: leave / re-enter bookkeeping
// The reference walks the agents sequentially (environment.py:832-864).  Everything position-only is done
// in parallel (one thread per agent; the agent x goal distance table lives in LDS); the occupancy / history
// vectors are walked in the reference's order: iteration i = [thread i runs observation(i)'s event] barrier
// [every thread a evaluates its row of graph_observation(i) on that snapshot] barrier [apply the rare
// "no free goal" clear].  The per-step assignment runs the lexifair group solver on 2^k-lane groups.
#pragma once
#include "fmarl_dev.h"
#include "fmarl_kernels.h"
#include "fmarl_lexifair.hip"
#include "fmarl_reset.hip"
#include "fmarl_step.hip"

namespace fmarl {

struct NavRow { float occ; int8_t code, hist; int16_t pad; };   // goal of an agent row: landmark index (-1 = own position), its
                                                              // occupancy and history (an agent index or -1): 8 bytes

struct FairNavLds {
    char *base, *dead;   // the env's block / its part of the second region (tables nobody reads once the emission starts; the
                         // emission windows alias that region)
    const Params &p;
    __device__ FairNavLds(const Params &p_, char *lds, uint32_t el)
        : base(lds + (size_t)el * p_.lds_env_bytes), dead(lds + p_.lds_stage + (size_t)el * p_.lds2_bytes), p(p_) {}
    __device__ double *stat() const { return (double *)(dead + p.lds_stat); }           // [pd_new | Dg_old | Dg_new | Tr_old | Tr_new] x N
    __device__ double2 *pos() const { return (double2 *)(base + p.lds_pos); }
    __device__ float4 *agentf() const { return (float4 *)(base + p.lds_agentf); }   // (vx, vy, newly-stopped, -)
    __device__ float2 *posf() const { return (float2 *)(base + p.lds_posf); }       // (float)pos of every entity
    __device__ const float4 *wallf() const { return (const float4 *)(base + p.lds_wallf); }   // (e0, axis + w/2, e1, axis - w/2)
    __device__ double *wall() const { return (double *)(base + p.lds_wall); }
    __device__ int *flag() const { return (int *)(base + p.lds_flag); }
    __device__ int *episode() const { return (int *)(base + p.lds_flag) + 2; }     // the env's episode counter (the key of its next placement)
    __device__ double *predraw() const { return (double *)dead; }                  // in-kernel reset: the first Philox blocks of the new episode (over stat / D / ...: dead by then)
    __device__ double *D() const { return (double *)(dead + p.n_D); }               // [N][L] |x_a - goal_g|
    __device__ double *minprox() const { return (double *)(dead + p.n_minprox); }   // [L] min_a |x_a - goal_g|
    __device__ double *occ() const { return (double *)(dead + p.n_occ); }           // [L] landmark_poses_occupied
    __device__ int8_t *hist() const { return (int8_t *)(dead + p.n_occ + p.L * 8); }   // [L] goal_history (agent indices: one byte each, as in the state)
    __device__ int *match() const { return (int *)(dead + p.n_match); }             // [N] goal_match_index (this step)
    __device__ NavRow *rows() const { return (NavRow *)(base + p.n_rows); }         // [N ego][N entity]
    __device__ int *words() const { return (int *)(dead + p.n_words); }             // a*, all-done, free-empty
    __device__ bool skip() const { return *flag() != 0; }

    // the 13 features of entity e in the row block of ego i (nf:1222-1334), ego part included:
    // [dv (2) | dx (2) | goal or dx (2) | occupancy, history / 1, index | dx or wall corners (4) | type]
    __device__ void node_row(uint32_t i, uint32_t e, float (&o)[13]) const {
        node_row_at(base, p.N, p.L, p.O, p.lds_posf, p.lds_agentf, p.lds_wallf, p.n_rows, i, e, o);
    }
    // (the table offsets as arguments: the emission loop hands in copies pinned in scalar registers)
    static __device__ __forceinline__ void node_row_at(const char *base, uint32_t N, uint32_t L, uint32_t O, uint32_t off_posf,
                                                       uint32_t off_agentf, uint32_t off_wallf, uint32_t off_rows, uint32_t i,
                                                       uint32_t e, float (&o)[13]) {
        const uint32_t first_obst = N + L, first_wall = first_obst + O;
        const float2 *posf_ = (const float2 *)(base + off_posf);
        const float4 *agentf_ = (const float4 *)(base + off_agentf);
        // differences of the f32 roundings (as the other two scenarios' rows): what a learner-side rebuild starts from
        const float2 pi = posf_[i], pe = posf_[e];
        const float dx = pe.x - pi.x, dy = pe.y - pi.y;
        // velocities as they stand when graph_observation(i) runs: reward(a <= i) may have stopped a
        const float4 ai = agentf_[i];
        const float vix = ai.z != 0.f ? 0.f : ai.x, viy = ai.z != 0.f ? 0.f : ai.y;
        float vex = 0.f, vey = 0.f;
        if (e < N) {
            const float4 ae = agentf_[e];
            const bool stopped = e <= i && ae.z != 0.f;
            vex = stopped ? 0.f : ae.x; vey = stopped ? 0.f : ae.y;
        }
        o[0] = vex - vix; o[1] = vey - viy;
        o[2] = dx; o[3] = dy; o[4] = dx; o[5] = dy; o[8] = dx; o[9] = dy; o[10] = dx; o[11] = dy;
        if (e < N) {
            const NavRow r = ((const NavRow *)(base + off_rows))[i * N + e];
            const float2 gl = r.code >= 0 ? posf_[N + (uint32_t)r.code] : pe;
            o[4] = gl.x - pi.x; o[5] = gl.y - pi.y;
            o[6] = r.occ; o[7] = (float)r.hist;
            o[12] = 0.f;
        } else {
            o[6] = 1.f;
            o[7] = e < first_obst ? (float)(e - N) : (e < first_wall ? 0.f : (float)(e - first_wall));
            o[12] = e < first_obst ? 1.f : (e < first_wall ? 2.f : 3.f);
        }
        if (e >= first_wall) {   // corners (e0, axis + w/2), (e1, axis - w/2)
            const float4 wc = ((const float4 *)(base + off_wallf))[e - first_wall];
            o[8] = wc.x - pi.x; o[9] = wc.y - pi.y; o[10] = wc.z - pi.x; o[11] = wc.w - pi.y;
        }
    }
};

// nf:592-613 is_obstacle_collision: obstacles at 2.0 (s + s), wall boxes padded by 1.5 s
__device__ __forceinline__ bool wall_box_hit_pad15(double2 x, double axis, double e0, double e1, int orient) {
    const double s = 1.5 * kEntitySize;
    const double pperp = orient == 0 ? x.y : x.x, ppar = orient == 0 ? x.x : x.y;
    return (axis - s <= pperp) && (pperp <= axis + s) && (e0 - s <= ppar) && (ppar <= e1 + s);
}

struct ObsGoal { int goal, second; double g_occ, g_hist, second_occ; };

// observation(i)'s walk over the occupancy / history vectors (nf:845-996); Drow = distances of agent i.
__device__ ObsGoal obs_event(const double *Drow, const double *minprox, double *occ, int8_t *hist, int L, int i,
                             double thr, double mod) {
    ObsGoal out;
    int c = 0, s2 = 0;
    double dmin = Drow[0], d2 = 1e300;
    for (int g = 1; g < L; ++g) {
        const double d = Drow[g];
        if (d < dmin) { d2 = dmin; s2 = c; dmin = d; c = g; }
        else if (d < d2) { d2 = d; s2 = g; }
    }
    out.second = s2; out.second_occ = occ[s2];
    if (dmin < mod) {
        int chosen = c, goal = c;
        for (int g = 0; g < L; ++g)
            if (Drow[g] < mod && occ[g] == 1.0 && !(minprox[g] < thr)) occ[g] = minprox[g];
        if (dmin < thr) { occ[chosen] = 1.0; hist[chosen] = (int8_t)i; }
        else {
            const double closest = minprox[chosen];
            if (occ[chosen] == 1.0) {
                if (closest < thr) {   // somebody sits on it: nearest goal that is not marked 1
                    int k = 0, kk = 0, best = -1;
                    double bd = 1e300;
                    for (int g = 0; g < L; ++g)
                        if (occ[g] != 1.0) { if (Drow[g] < bd) { bd = Drow[g]; best = g; k = kk; } ++kk; }
                    if (best >= 0) { goal = best; chosen = k; }   // quirk nf:888-902: the COMPACT index is used below
                } else occ[chosen] = 1.0 - closest;
            } else occ[chosen] = 1.0 - closest;
        }
        out.goal = goal; out.g_occ = occ[chosen]; out.g_hist = (double)hist[chosen];
    } else {
        int best = -1;
        double bd = 1e300;
        for (int g = 0; g < L; ++g)
            if (occ[g] != 1.0 && Drow[g] < bd) { bd = Drow[g]; best = g; }
        if (best >= 0) { out.goal = best; out.g_occ = occ[best]; out.g_hist = (double)hist[best]; }
        else {
            for (int g = 0; g < L; ++g) occ[g] = 0.0;
            out.goal = -1; out.g_occ = 0.0; out.g_hist = (double)hist[i];
        }
    }
    return out;
}

// N <= 3 (the shipped FA+FR configuration): the lexicographic-fair assignment by enumeration -- the six permutations'
// key vectors (key = (cost, row * G + col), the total order lexifair_group uses), each sorted descending, compared
// lexicographically; every lane of the group evaluates all of them from the env's LDS cost table (broadcast reads) and
// keeps its own row's column.  Same result as lexifair_group (the optimum under a total order is unique), a third of its
// instructions at N = 3.  Rows / columns beyond N carry the same dummy key in every admissible permutation.
template <int G>
__device__ __forceinline__ int lexifair_upto3(const double *D, int L, int N, int lane) {
    double c[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[r][j] = (r < N && j < N) ? D[r * L + j] : -1e300;
    constexpr int P[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    double bk0 = 0, bk1 = 0, bk2 = 0;
    int bi0 = 0, bi1 = 0, bi2 = 0, best = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const bool admissible = N >= 3 || (N == 2 && P[q][2] == 2) || q == 0;
        double k0 = c[0][P[q][0]], k1 = c[1][P[q][1]], k2 = c[2][P[q][2]];
        int i0 = P[q][0], i1 = G + P[q][1], i2 = 2 * G + P[q][2];
        // sort the three keys descending: compare-exchange (0,1), (1,2), (0,1)
#define FMARL_CX(ka, ia, kb, ib) { const bool sw = ka < kb || (ka == kb && ia < ib); const double tk = sw ? kb : ka; const int ti = sw ? ib : ia; \
                                   kb = sw ? ka : kb; ib = sw ? ia : ib; ka = tk; ia = ti; }
        FMARL_CX(k0, i0, k1, i1) FMARL_CX(k1, i1, k2, i2) FMARL_CX(k0, i0, k1, i1)
#undef FMARL_CX
        const bool lt0 = k0 < bk0 || (k0 == bk0 && i0 < bi0), eq0 = k0 == bk0 && i0 == bi0;
        const bool lt1 = k1 < bk1 || (k1 == bk1 && i1 < bi1), eq1 = k1 == bk1 && i1 == bi1;
        const bool lt2 = k2 < bk2 || (k2 == bk2 && i2 < bi2);
        const bool take = q == 0 || (admissible && (lt0 || (eq0 && (lt1 || (eq1 && lt2)))));
        bk0 = take ? k0 : bk0; bk1 = take ? k1 : bk1; bk2 = take ? k2 : bk2;
        bi0 = take ? i0 : bi0; bi1 = take ? i1 : bi1; bi2 = take ? i2 : bi2;
        best = take ? q : best;
    }
    // column of row `lane` in permutation `best`: 2 bits per row, 6 bits per permutation
    constexpr unsigned long long codes = (0ull | 1ull << 2 | 2ull << 4) | (0ull | 2ull << 2 | 1ull << 4) << 6 | (1ull | 0ull << 2 | 2ull << 4) << 12 |
                                         (1ull | 2ull << 2 | 0ull << 4) << 18 | (2ull | 0ull << 2 | 1ull << 4) << 24 | (2ull | 1ull << 2 | 0ull << 4) << 30;
    return (int)((codes >> (6 * best + 2 * (lane < 3 ? lane : 0))) & 3ull);
}

// `reset_match`: the assignment of the envs that were reset inside this step (marked in words()[2]), written straight into the state
// (goal_match of the workgroup's first env): nothing in the step reads it -- no table, no barrier behind it.
template <int G, int THREADS>
__device__ __forceinline__ void fairnav_assign_tasks(const Params &p, char *lds, int nenv, int *reset_match, bool warm = false) {
    const int group = threadIdx.x / G, ngroups = THREADS / G, lane = threadIdx.x % G;
    for (int el = group; el < nenv; el += ngroups) {
        const FairNavLds t(p, lds, el);
        if (reset_match && t.words()[2] == 0) continue;   // (group-uniform)
        int *out = reset_match ? reset_match + (size_t)el * p.N : t.match();
        if (G == 4 && p.N <= 3) {
            const int mc = lexifair_upto3<G>(t.D(), p.L, p.N, lane);
            if (lane < p.N) out[lane] = mc;
            continue;
        }
        double c[G];
#pragma unroll
        for (int j = 0; j < G; ++j) c[j] = (lane < p.N && j < p.N) ? t.D()[lane * p.L + j] : 0.0;
        // `warm`: the step's re-assignment starts from the previous step's, which the env's match table holds (a launch that loaded the
        // state loaded it too; inside a span the emission windows have been over it: the greedy start); so does a freshly placed env's first
        const int mc = lexifair_group<G>(c, p.N, (warm && lane < p.N) ? t.match()[lane] : -1);
        if (lane < p.N) out[lane] = mc;
    }
}

// node_obs rows of the workgroup's envs (nf:1222-1334) from the LDS tables: shared by the step / reset passes and the
// learner-side rebuild (fairnav_rebuild_kernel).
// ALL_ROWS (the step pass): envs that ended in this step and are reset by the same launch get their rows written as well --
// the reset pass that follows overwrites them -- so that one ended env does not send the whole workgroup down the per-lane
// path (13 four-byte stores per row at a 52-byte stride: half the store rate; with episodes ending at all phases nearly every
// workgroup has such an env in every step)
template <bool ALL_ROWS, int THREADS>
__device__ __forceinline__ void fairnav_emit_rows(const Params &p, const FmarlOutputs &o, char *lds, int env0, int nenv) {
    const int tid = threadIdx.x, N = p.N;
    if (o.node_obs) {
        // one lane per (ego, entity) row: the 13 features share their loads; the rows leave through the waves' LDS
        // windows (fmarl_step.hip flush_rows) unless some env of the workgroup keeps its previous rows
        const uint32_t NE = N * p.E, total = nenv * NE;
        float *dst = o.node_obs + (size_t)env0 * NE * 13;
        // (the barrier is also what separates the last readers of the second region's tables from the windows that alias it)
        const bool any_skip = __syncthreads_or(tid < nenv && FairNavLds(p, lds, tid).skip());
        const bool some_skip = !ALL_ROWS && any_skip;
        if (!some_skip) {
            // what the loop needs of the kernel arguments, pinned (fmarl_dev.h pin_sgpr)
            const uint32_t kN = pin_sgpr((uint32_t)p.N), kL = pin_sgpr((uint32_t)p.L), kO = pin_sgpr((uint32_t)p.O), kE = pin_sgpr((uint32_t)p.E);
            const uint32_t k_env = pin_sgpr((uint32_t)p.lds_env_bytes), k_posf = pin_sgpr((uint32_t)p.lds_posf), k_agentf = pin_sgpr((uint32_t)p.lds_agentf);
            const uint32_t k_wallf = pin_sgpr((uint32_t)p.lds_wallf), k_rows = pin_sgpr((uint32_t)p.n_rows);
            FastDiv dNE, dE;
            dNE.m = pin_sgpr(p.dC4.m); dNE.d = p.dC4.d; dE.m = pin_sgpr(p.dE.m); dE.d = p.dE.d;
            for (uint32_t base = 0; base < total; base += THREADS) {
                const uint32_t q = base + tid, w0 = base + (tid & ~63u);
                float row[13];
                if (q < total) {
                    const uint32_t e_l = dNE.div(q), r = q - e_l * NE, a = dE.div(r), e = r - a * kE;   // dC4 = N * E
                    FairNavLds::node_row_at(lds + (size_t)e_l * k_env, kN, kL, kO, k_posf, k_agentf, k_wallf, k_rows, a, e, row);
                }
                flush_rows<13, true>(p, lds, row, w0 < total ? (int)min(64u, total - w0) : 0, dst + (size_t)w0 * 13);
            }
        } else {
            for (uint32_t q = tid; q < total; q += THREADS) {
                const uint32_t e_l = p.dC4.div(q);
                const FairNavLds te(p, lds, e_l);
                if (te.skip()) continue;
                const uint32_t r = q - e_l * NE, a = p.dE.div(r), e = r - a * p.E;
                float row[13];
                te.node_row(a, e, row);
#pragma unroll
                for (int f = 0; f < 13; ++f) dst[(size_t)q * 13 + f] = row[f];
            }
        }
    }
}

// Placement table of the in-kernel reset (fmarl_reset.hip place_env): the env's own LDS entity table in float64.
struct PlacedEnvLds {
    double2 *pos;
    const Params &p;
    int env;
    __device__ double2 g_obstacle(int k) const { return pos[p.N + p.L + k]; }
    __device__ double2 g_agent(int k) const { return pos[k]; }
    __device__ double2 g_landmark(int k) const { return pos[p.N + k]; }
    __device__ void set_obstacle(int k, double2 x) { pos[p.N + p.L + k] = x; p.obstacle_pos[(size_t)env * p.O + k] = x; }
    __device__ void set_agent(int k, double2 x) { pos[k] = x; p.agent_pos[(size_t)env * p.N + k] = x; }
    __device__ void set_landmark(int k, double2 x) { pos[p.N + k] = x; p.landmark_pos[(size_t)env * p.L + k] = x; }
    // The placing lane is alone in its wave and every LDS read is a round trip it waits for (a test per trip: 15 000 cycles per
    // placement at three agents, profiles/r5_notes.md): four table entries are read per trip (the last one repeated past the end:
    // harmless for an OR) and tested on the squared distance; only a pair within an ulp of the threshold takes the square root.
    __device__ bool any_closer(int kind, int k, double2 x, double thr) const {
        const double2 *tab = kind == 0 ? pos + p.N + p.L : (kind == 1 ? pos : pos + p.N);
        const double c2 = thr * thr, lo = c2 * (1.0 - 1e-15), hi = c2 * (1.0 + 1e-15);
        bool hit = false;
        for (int j0 = 0; j0 < k; j0 += 4) {
            const int last = k - 1;
            const double2 q0 = tab[j0], q1 = tab[min(j0 + 1, last)], q2 = tab[min(j0 + 2, last)], q3 = tab[min(j0 + 3, last)];
            const double s0 = (q0.x - x.x) * (q0.x - x.x) + (q0.y - x.y) * (q0.y - x.y), s1 = (q1.x - x.x) * (q1.x - x.x) + (q1.y - x.y) * (q1.y - x.y);
            const double s2 = (q2.x - x.x) * (q2.x - x.x) + (q2.y - x.y) * (q2.y - x.y), s3 = (q3.x - x.x) * (q3.x - x.x) + (q3.y - x.y) * (q3.y - x.y);
            hit |= (s0 < lo) | (s1 < lo) | (s2 < lo) | (s3 < lo);
            const bool band = (!(s0 < lo) & !(s0 > hi)) | (!(s1 < lo) & !(s1 > hi)) | (!(s2 < lo) & !(s2 > hi)) | (!(s3 < lo) & !(s3 > hi));
            if (band) hit |= closer_than(q0, x, thr) | closer_than(q1, x, thr) | closer_than(q2, x, thr) | closer_than(q3, x, thr);
        }
        return hit;
    }
};

// The in-kernel reset's placement by TEAMS of lanes (round 6).  One lane per ended env walked the reference's rejection sampling
// alone -- every distance test a dependent LDS round trip, 8 900 cycles per placement at three agents while the other 191 lanes of
// the workgroup waited at the barrier (profiles/r5_ticks_fnav_steady.txt).  Here every ended env gets a team of 16 (32, 64) lanes, all
// of them idle in this phase anyway: lane j of the team HOLDS entity j (placement order: obstacles, agents, goals) in registers.  The
// sequence stays the reference's -- one candidate per Philox block, in stream order, accepted or rejected before the next one is
// looked at, so the stream consumption and every decision are the sequential walk's, bit for bit -- but a candidate's tests run side
// by side: each lane tests the candidate against the entity it holds (the threshold of its kind), one ballot says whether any hit,
// and the lane of the slot being filled takes the candidate.  A trip of the loop is a dozen vector instructions and a ballot; the
// next block's LDS read is issued a trip ahead.  Which team takes which env comes from the envs' flags (ballots every wave takes for
// itself: no worklist, no extra barrier).  Entities beyond 64 per env: the one-lane walk (place_env).
__device__ __forceinline__ double u_lin(double lo, double hi, double a) { return lo + (hi - lo) * a; }   // PhiloxStream::uniform's expression
template <int THREADS, int NL>
__device__ __forceinline__ void fairnav_place_teams(const Params &p, char *lds, int env0, int nenv, int n_pre) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int O = NL ? 3 : p.O, N = NL ? NL : p.N, L = NL ? NL : p.L, W = NL ? 0 : p.W, slots = O + N + L;
    const int ts_log = slots <= 16 ? 4 : (slots <= 32 ? 5 : 6), TS = 1 << ts_log;
    const int j = tid & (TS - 1), team = tid >> ts_log, nteams = THREADS >> ts_log;
    const uint64_t tmask = TS == 64 ? ~0ull : ((1ull << TS) - 1ull);
    const int tshift = lane & ~(TS - 1);
    // the ended envs (flag 0: fairnav_pass), every wave for itself
    const uint64_t m0 = __ballot(lane < nenv && !FairNavLds(p, lds, lane).skip());
    const uint64_t m1 = nenv > 64 ? __ballot(lane + 64 < nenv && !FairNavLds(p, lds, lane + 64).skip()) : 0ull;
    const int c0 = __popcll(m0), n_ended = c0 + __popcll(m1);
    const double ws = p.world_size;
    const double thr = 1.05 * (kEntitySize + kEntitySize), thr_goal = 1.2 * (kEntitySize + kEntitySize);   // nf:643
    const double thr_obs = 2.0 * (kEntitySize + kEntitySize);                                              // nf:592-613
    for (int first = 0; first < n_ended; first += nteams) {
        const int kth = first + team;
        if (kth >= n_ended) continue;   // (team-uniform; the ballots below only count the lanes that are here)
        uint64_t m = kth < c0 ? m0 : m1;
        for (int q = kth < c0 ? kth : kth - c0; q > 0; --q) m &= m - 1;
        const int el = (kth < c0 ? 0 : 64) + __ffsll((unsigned long long)m) - 1;
        const FairNavLds te(p, lds, el);
        const int env = env0 + el, epi = *te.episode();
        const uint32_t genv = (uint32_t)(p.env_offset + env);
        // the first blocks of the env's stream: lane j of the team draws block j into the env's part of the second LDS region (its
        // tables there are dead by now); the team's lanes share a wave, so a wavefront-scope fence is all that orders write and reads
        double *pre = te.predraw();
        const int n_team = n_pre < TS ? n_pre : TS;
        double a, b;
        if (j < n_team) {
            philox_block(p.seed, (uint32_t)j, genv, (uint32_t)epi, a, b);
            pre[2 * j] = a; pre[2 * j + 1] = b;
        }
        wave_sync();
        auto block = [&](int idx, double &a, double &b) {
            if (idx < n_team) { a = pre[2 * idx]; b = pre[2 * idx + 1]; }
            else philox_block(p.seed, (uint32_t)idx, genv, (uint32_t)epi, a, b);
        };
        block(0, a, b);
        const double wlen = u_lin(0.2, 0.8, a) * p.world_size / 4;   // nf:239-241
        double2 mine = make_double2(0.0, 0.0);
        if (j < O) {   // nf:271-275, every obstacle by its own lane
            block(1 + j, a, b);
            mine = make_double2(0.8 * u_lin(-ws / 2, ws / 2, a), 0.8 * u_lin(-ws / 2, ws / 2, b));
        }
        block(1 + O, a, b);
        const double wall_position = u_lin(0.2, 0.9, a);   // :288, drawn even without walls
        double axis0 = 0.0, axis1 = 0.0;   // :294-324 (W <= 2; scalars, not arrays: a run-time index would put them in scratch memory)
        int orient0 = 0, orient1 = 0;
        if (W > 0) { block(2 + O, a, b); orient0 = a < 0.5 ? 0 : 1; axis0 = wall_position * ws / 2; }
        if (W > 1) { block(3 + O, a, b); orient1 = a < 0.5 ? 0 : 1; axis1 = -wall_position * ws / 2; }
        int idx = 2 + O + W, e = O, tries = 0, fails = 0;
        double an, bn;
        block(idx, a, b);
        block(idx + 1, an, bn);
        while (e < slots) {   // :389-457 agents, :472-535 goals
            const bool goal = e >= O + N;
            double2 x = make_double2(u_lin(-ws / 2, ws / 2, a), u_lin(-ws / 2, ws / 2, b));
            if (goal) x = make_double2(0.8 * x.x, 0.8 * x.y);
            const bool relevant = j < O || (j >= (goal ? O + N : O) && j < e);
            bool hit = relevant && closer_than(mine, x, j < O ? thr_obs : (goal ? thr_goal : thr));
            if (W > 0) hit |= wall_box_hit_pad15(x, axis0, -wlen, wlen, orient0);
            if (W > 1) hit |= wall_box_hit_pad15(x, axis1, -wlen, wlen, orient1);
            const bool bad = ((__ballot(hit) >> tshift) & tmask) != 0;
            ++tries;
            if (!bad || tries >= kMaxTries) {
                fails += bad ? 1 : 0;
                if (j == e) mine = x;
                ++e; tries = 0;
            }
            ++idx;
            a = an; b = bn;
            block(idx + 1, an, bn);
        }
        // every lane stores the entity it holds: into the env's float64 table (what the re-seated lanes read) and into the state
        if (j < O) { te.pos()[N + L + j] = mine; p.obstacle_pos[(size_t)env * O + j] = mine; }
        else if (j < O + N) te.pos()[j - O] = mine;   // (into the state by the agent's own lane when it re-seats: behind its terminal step's store)
        else if (j < slots) { te.pos()[N + (j - O - N)] = mine; p.landmark_pos[(size_t)env * L + (j - O - N)] = mine; }
        if (j == 0) {
            p.wall_length[env] = wlen;
            const size_t g = (size_t)env * W;
            if (W > 0) { p.wall_orient[g] = orient0; p.wall_axis[g] = axis0; p.wall_e0[g] = -wlen; p.wall_e1[g] = wlen; }
            if (W > 1) { p.wall_orient[g + 1] = orient1; p.wall_axis[g + 1] = axis1; p.wall_e0[g + 1] = -wlen; p.wall_e1[g + 1] = wlen; }
            p.place_fails[env] = fails;
            p.episode[env] = epi + 1;
            *te.episode() = epi + 1;
        }
    }
}

// One pass over the workgroup's envs.
//   STEP = true : one env step (MultiAgentGraphEnv.step, environment.py:816-877).  Envs whose agents are all done (stop-on-goal
//                 `status` or the episode length) are reset INSIDE the pass when auto_reset is set (the vec-env worker's
//                 behaviour, env_wrappers.py:859-865; episodes of this scenario end env by env, at any step: a reset pipeline
//                 of its own would have to be launched behind every step -- three launches that nearly always find nothing to
//                 do, 31 % of the step time at 65 536 x 3).  Everything the terminal step still owes -- reward, done, info, the
//                 counters -- is computed BEFORE the sequential occupancy walk, which none of it depends on; then the ended
//                 envs are placed (one lane per env, on the env's LDS entity table), their lanes re-seated on the new episode
//                 (distance table, fair assignment nf:469), and the walk, the observation and the emission run ONCE for all
//                 envs -- the ended ones on their new state.  (Rounds 2-3 ran a complete second pass over the workgroup for
//                 the ended envs: a second dependent chain of a dozen barriers behind the first one in nearly every
//                 workgroup of a training run, where episodes end at all phases -- 0.052 ms per launch when no env ends,
//                 0.086 when two or three per workgroup do.)
//   STEP = false: the observation part of an explicit reset (fmarl_reset: placement and assignment by their own kernels).
//   `carry` (span kernel): bit 0 = this agent's state arrives in `c` (left there by the previous step of the span), the static
//                 entities of the workgroup's envs and their episode counters are still in the LDS tables -- nothing is loaded but the
//                 action; bit 1 = the new state stays in `c` instead of going to global memory.  0 = a step of its own.
//   THREADS       workgroup size: 192 when the envs' agent lanes fit three waves (the shipped FA+FR configuration: 64 envs x 3 agents)
//                 -- a fourth wave would hold no agent, and without it the kernel has 168 vector registers per lane instead of 128:
//                 what the state needs to stay in registers across a span (round 3 needed 181 for it at 256 threads and spilled).
// (small integers packed: `bits` = status | (goal_history + 1) << 1 | (goal_reached + 1) << 9 | step << 17, `hits` = obstacle | agent << 16
// collision counts -- a byte-sized field would still take a whole register; min_time is constant over an episode and re-read)
struct FairnavCarry { double2 x, v; double pd, Dg, Tr, left, occ; uint32_t bits, hits; };
__device__ __forceinline__ uint32_t fairnav_pack(double status, int hist, double gr, int step) {
    return (status != 0.0 ? 1u : 0u) | ((uint32_t)(hist + 1) & 0xffu) << 1 | ((uint32_t)((int)gr + 1) & 0xffu) << 9 | (uint32_t)step << 17;
}

//   NL            0, or 3: the shipped FA / FA+FR configuration's shape -- 3 agents, 3 goals, 3 obstacles, no wall -- as compile-time
//                 constants.  The pass is a chain of short loops over agents / goals / partners whose bodies start with an LDS read:
//                 with run-time bounds every trip waits for its own read; with the bounds known the loops unroll and a loop's reads
//                 leave together (round 6: one launch per step 0.0527 -> 0.0483 ms, the span 0.0431 -> 0.0403, 65 536 envs).
template <bool STEP, int THREADS, int NL = 0>
__device__ __forceinline__ void fairnav_pass(const Params &p, const FmarlOutputs &o, char *lds, const int32_t *action_idx,
                                             const float *action_vec, int auto_reset, FairnavCarry &c, const int carry) {
    FMARL_TICKS_BEGIN
    const int tid = threadIdx.x, N = NL ? NL : p.N, L = NL ? NL : p.L, O = NL ? 3 : p.O, W = NL ? 0 : p.W;
    const int env0 = env_block(p) * p.epb;
    const int nenv = min(p.epb, p.n_envs - env0);
    const int el = tid / N, i = tid - el * N;
    const bool in_range = el < nenv;
    const int env = env0 + el;
    const size_t g = (size_t)env * N + i;
    const FairNavLds t(p, lds, el);
    double *s_stat = t.stat();
    const bool flagged = STEP ? false : (in_range && p.reset_flag[env] != 0);
    const bool active = in_range;
    const bool arrives = STEP && (carry & 1) != 0, keep = STEP && (carry & 2) != 0;

    double2 x = make_double2(0, 0), v = make_double2(0, 0);
    double pd = 0, status = 0, occ_i = 0, mtime = 0;   // (mtime: written by the in-kernel reset, read by the info planes)
    double Dg_old = 0, Tr_old = 0, left = 0, gr = 0;
    int step = 0, hist_i = 0, noc_old = 0, nac_old = 0;
    if (arrives) {
        x = c.x; v = c.v; pd = c.pd; occ_i = c.occ;
        status = (double)(c.bits & 1u); hist_i = (int)((c.bits >> 1) & 0xffu) - 1; gr = (double)((int)((c.bits >> 9) & 0xffu) - 1); step = (int)(c.bits >> 17) + 1;
        Dg_old = c.Dg; Tr_old = c.Tr; left = c.left; noc_old = (int)(c.hits & 0xffffu); nac_old = (int)(c.hits >> 16);
    } else if (active) {
        x = p.agent_pos[g]; v = p.agent_vel[g]; pd = p.p_dist[g]; status = (double)p.status[g];
        occ_i = p.goal_occ[g]; hist_i = p.goal_history[g];   // L == N
        step = p.cur_step[env] + (STEP ? 1 : 0);
        if (STEP && i == 0) *t.episode() = p.episode[env];
        if (STEP && N > 3) t.match()[i] = p.goal_match[g];   // the previous step's assignment: where this step's solve starts (lexifair_group `warm`)
    }
    if (STEP && active && o.info) mtime = p.min_time[g];   // (constant over an episode: a cached load, issued here so that nothing waits for it)
    // the agent's action index: a load from the tape in HBM that the contact forces would otherwise wait for at the head of the physics
    const int a_pre = (STEP && active && action_idx) ? action_idx[g] : -1;
    if (active) {
        t.pos()[i] = x;
        t.occ()[i] = occ_i; t.hist()[i] = (int8_t)hist_i;
        if (i == 0) { t.words()[0] = N; t.words()[1] = 1; t.words()[2] = 0; }
    }
    // reset observation: workgroups without a freshly reset env have nothing to do (block-uniform exit)
    if (!STEP && !__syncthreads_or(flagged)) return;
    if (in_range && i == 0) *t.flag() = (STEP || flagged) ? 0 : 1;   // (a step emits every env: the ended ones after their reset)
    if (!arrives) load_statics_range(p, lds, env0, 0, nenv, tid, THREADS);
    __syncthreads();
    FMARL_TICK(0);   // state loads, entity tables, barrier
    if (STEP && active) world_step_agent<NL, NL ? 3 : 0, 0>(p, t.base, i, g, action_idx, action_vec, x, v, pd, status == 0.0, a_pre);
    __syncthreads();   // every lane has finished reading the old positions
    FMARL_TICK(1);   // physics
    if (active) t.pos()[i] = x;
    __syncthreads();

    if (active) {   // distance table and per-goal minimum over agents
        for (int k = 0; k < L; ++k) t.D()[i * L + k] = dist2(x, t.pos()[N + k]);
        double m = 1e300;
        for (int a = 0; a < N; ++a) m = fmin(m, dist2(t.pos()[a], t.pos()[N + i]));
        t.minprox()[i] = m;
    }
    __syncthreads();
    FMARL_TICK(2);   // distance table
    if (STEP && !FMARL_SKIP(p, 64)) {
        // reward(agent 0)'s lexicographic-fair re-assignment on the new positions (nf:704-721)
        if (N <= 4) fairnav_assign_tasks<4, THREADS>(p, lds, nenv, nullptr, !arrives);
        else if (N <= 8) fairnav_assign_tasks<8, THREADS>(p, lds, nenv, nullptr, !arrives);
        else if (N <= 16) fairnav_assign_tasks<16, THREADS>(p, lds, nenv, nullptr, !arrives);
        else fairnav_assign_tasks<32, THREADS>(p, lds, nenv, nullptr, !arrives);
    } else if (active) {
        t.match()[i] = p.goal_match[g];
    }
    __syncthreads();
    FMARL_TICK(3);   // assignment

    // reward's status transition (nf:726-741) and the per-agent info bookkeeping (nf:489-573) only need
    // positions and the agent's own previous values
    double Dg_new = 0, Tr_new = 0, dgoal = 0;
    bool newly = false, done = false;
    if (active) {
        if (STEP) {
            dgoal = t.D()[i * L + t.match()[i]];
            newly = dgoal < p.thr && status == 0.0;
            if (newly) status = 1.0;
            done = status != 0.0 || step >= p.episode_length;   // environment.py:237-247
            if (!done) atomicAnd(&t.words()[1], 0);
            if (!arrives) { Dg_old = p.dists_to_goal[g]; Tr_old = p.times_required[g]; left = p.dist_left[g]; gr = (double)p.goal_reached[g]; }
            int near = 0;
            double dn = t.D()[i * L];
            for (int k = 1; k < L; ++k) { const double d = t.D()[i * L + k]; if (d < dn) { dn = d; near = k; } }
            Dg_new = Dg_old; Tr_new = Tr_old;
            const double now = step * kDt;
            if (dn < p.thr && ((double)near != gr && gr != -1.0)) { gr = near; left = dn; }
            if (dn < p.thr && Tr_new == -1.0) { Tr_new = now; Dg_new = pd; left = dn; gr = near; }
            if (Tr_new == -1.0) { Dg_new = pd; left = dn; }
            if (dn > p.thr && Tr_new != -1.0) { Dg_new = pd; Tr_new = now; left = dn; }
            if (dn < p.thr && (double)near == gr) { left = dn; gr = near; }
            s_stat[i] = pd; s_stat[N + i] = Dg_old; s_stat[2 * N + i] = Dg_new;
            s_stat[3 * N + i] = Tr_old; s_stat[4 * N + i] = Tr_new;
        }
        t.agentf()[i] = make_float4((float)v.x, (float)v.y, newly ? 1.f : 0.f, 0.f);
        t.posf()[i] = make_float2((float)x.x, (float)x.y);
    }
    __syncthreads();
    const bool ended = STEP && active && auto_reset && t.words()[1] != 0;   // every agent of the env is done: it is reset below
    const bool emit = STEP ? true : flagged;
    FMARL_TICK(4);   // status transition, bookkeeping, barrier

    if (STEP) {
        if (active) {   // what the step owes besides the observation: reward, done, info, the state -- none of it depends on the walk
            double fairness, m, sd;   // nf:693-698, same stale / fresh rule as navigation_graph
            if (Dg_old == -1.0) mixed_stats(s_stat, s_stat, N, N, m, sd);
            else mixed_stats(s_stat + 2 * N, s_stat + N, N, i, m, sd);
            fairness = ratio_out(m, sd + 0.0001);
            int ag_hits = 0;
            for (int j = 0; j < N; ++j)
                if (j != i && closer_than(x, t.pos()[j], 1.05 * (kEntitySize + kEntitySize))) ++ag_hits;
            bool ob_hit = false;
            for (int k = 0; k < O; ++k)
                ob_hit |= closer_than(t.pos()[N + L + k], x, 2.0 * (kEntitySize + kEntitySize));
            const double *wl = t.wall();
            for (int w = 0; w < W; ++w)
                ob_hit |= wall_box_hit_pad15(x, wl[w * 4], wl[w * 4 + 1], wl[w * 4 + 2], (int)wl[w * 4 + 3]);
            double rew = 0.0;
            if (dgoal < p.thr) { if (newly) rew += p.goal_rew; }
            else rew -= dgoal;
            rew -= p.collision_rew * ag_hits;
            if (ob_hit) rew -= p.collision_rew;
            double fr = p.fair_rew * tanh_out(fairness - p.zeroshift);
            if (fr < -p.fair_rew) fr = -p.fair_rew;
            rew = fmin(fmax(rew + fr, -2 * p.collision_rew), p.goal_rew + p.fair_rew);

            if (!arrives) { noc_old = p.num_obst_coll[g]; nac_old = p.num_agent_coll[g]; }
            const int noc = noc_old + (ob_hit ? 1 : 0), nac = nac_old + ag_hits;
            const double2 vout = newly ? make_double2(0.0, 0.0) : v;   // nf:736-737
            if (keep) {
                c.x = x; c.v = vout; c.pd = pd; c.Dg = Dg_new; c.Tr = Tr_new; c.left = left;
                c.bits = fairnav_pack(status, 0, gr, step);   // (the history entry follows the walk)
                c.hits = (uint32_t)noc | (uint32_t)nac << 16;
            } else {
                p.agent_pos[g] = x; p.agent_vel[g] = vout; p.p_dist[g] = pd; p.status[g] = (int8_t)status;
                p.dists_to_goal[g] = Dg_new; p.times_required[g] = Tr_new; p.dist_left[g] = left; p.goal_reached[g] = (int8_t)gr;
                p.num_obst_coll[g] = noc; p.num_agent_coll[g] = nac;
                if (!ended) p.goal_match[g] = t.match()[i];   // (an env that is reset below: the new episode's assignment, by the lanes that solve it)
                if (i == 0) p.cur_step[env] = step;
            }
            if (o.reward) o.reward[g] = (float)rew;
            if (o.done) o.done[g] = done;
            if (o.info) {   // nf:573-590
                double dm, ds, tm, ts;
                mixed_stats(s_stat + 2 * N, s_stat + N, N, i + 1, dm, ds);
                mixed_stats(s_stat + 4 * N, s_stat + 3 * N, N, i + 1, tm, ts);
                const size_t plane = (size_t)p.n_envs * N;
                float *inf = o.info + g;
                inf[FMARL_INFO_DIST_TO_GOAL * plane] = (float)left;
                inf[FMARL_INFO_TIME_REQ_TO_GOAL * plane] = (float)Tr_new;
                inf[FMARL_INFO_NUM_AGENT_COLLISIONS * plane] = (float)nac;
                inf[FMARL_INFO_NUM_OBST_COLLISIONS * plane] = (float)noc;
                inf[FMARL_INFO_DISTANCE_MEAN * plane] = (float)dm;
                inf[FMARL_INFO_DISTANCE_VARIANCE * plane] = (float)ds;
                inf[FMARL_INFO_MEAN_BY_VARIANCE * plane] = (float)ratio_out(dm, ds + 0.0001);
                inf[FMARL_INFO_DISTS_TRAVELED * plane] = (float)Dg_new;
                inf[FMARL_INFO_TIME_TAKEN * plane] = (float)Tr_new;   // nf:582: times_required again
                inf[FMARL_INFO_TIME_MEAN * plane] = (float)tm;
                inf[FMARL_INFO_TIME_STDDEV * plane] = (float)ts;
                inf[FMARL_INFO_TIME_MEAN_BY_STDDEV * plane] = (float)ratio_out(tm, ts + 0.0001);
                inf[FMARL_INFO_MIN_TIME_TO_GOAL * plane] = (float)mtime;
                inf[FMARL_INFO_INDIVIDUAL_REWARD * plane] = (float)rew;
            }
        }
        FMARL_TICK(6);   // reward, statistics, state stores, info planes
        // ---- the envs that ended: the reset of env_wrappers.py:859-865, in place.  (The barrier also separates the terminal step's
        // state stores and table reads above from the placement's stores to the same fields and its writes to the tables.)
        // (flag: marks the envs the pre-draw and the restricted assignment below work on; written here so that the barrier that
        // decides whether any env ended also publishes it -- nothing between the previous barrier and this point reads the flags)
        if (in_range && i == 0 && auto_reset) { *t.flag() = ended ? 0 : 1; t.words()[2] = ended ? 1 : 0; }
        if (__syncthreads_or(ended)) {
            if (in_range && i == 0) p.reset_flag[env] = ended ? 1 : 0;
            // The placement is the reference's sequential rejection sampling on the env's own Philox stream, by ONE lane per ended
            // env while the workgroup waits: eleven or more Philox blocks of ten dependent rounds each were 14 % of a wave's cycles
            // with episodes ending at all phases (profiles/r4_ticks_fnav_steady.txt).  The blocks are counter-based, so the first
            // p.n_pre of them are drawn here by all lanes side by side (one block each) into the env's part of the second LDS
            // region (its tables there are dead by now); the placing lane then only reads them and runs the distance tests.
            // (Round 6: where teams place -- fairnav_place_teams -- a team draws its env's blocks itself and the workgroup's pre-draw pass
            // with its barrier is gone; the lanes of a team share a wave, LDS keeps a wave's accesses in order.)
            const int n_pre = p.n_pre;
            const bool teams = O + N + L <= 64;
            if (!teams) {
                for (int task = tid; task < nenv * n_pre; task += THREADS) {
                    const int e_l = task / n_pre, b = task - e_l * n_pre;
                    const FairNavLds te(p, lds, e_l);
                    if (te.skip()) continue;
                    double u0, u1;
                    philox_block(p.seed, (uint32_t)b, (uint32_t)(p.env_offset + env0 + e_l), (uint32_t)*te.episode(), u0, u1);
                    te.predraw()[2 * b] = u0; te.predraw()[2 * b + 1] = u1;
                }
                __syncthreads();
            }
            FMARL_TICK(11);   // (measure builds) the barrier that found ended envs (+ the pre-draw pass of the one-lane walk)
            if (teams) fairnav_place_teams<THREADS, NL>(p, lds, env0, nenv, n_pre);   // a team of lanes per ended env
            else if (in_range && i == 0 && ended) {   // more than 64 entities: the first lane of the env walks alone
                PlacedEnvLds pl{t.pos(), p, env};
                const int epi = *t.episode();
                place_env(p, pl, kResetAuto, env, kPlaceEntities, epi, t.predraw(), n_pre);
                *t.episode() = epi + 1;
            }
            FMARL_TICK(12);   // (measure builds) the placement itself
            __threadfence_block();   // (walls: the static entities of the placed envs are re-read from the state below)
            __syncthreads();
            FMARL_TICK(13);   // (measure builds) the barrier behind the placement: a wave without a team waits here for the slowest one
            if (ended) {   // re-seat the env's lanes on the new episode (reset_world: nf:233-241, environment.py:882-898)
                if (W == 0) {   // the placement left the new landmarks / obstacles in the env's float64 table: their f32 copies from there
                    for (int k = i; k < L + O; k += N) { const double2 e = t.pos()[N + k]; t.posf()[N + k] = make_float2((float)e.x, (float)e.y); }
                } else load_statics_range(p, lds, env0, el, el + 1, i, N);   // (+ the wall tables: a round trip through global memory)
                const double2 nx = t.pos()[i];   // (the placement wrote the env's float64 table)
                t.occ()[i] = 0.0; t.hist()[i] = -1;
                t.agentf()[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                t.posf()[i] = make_float2((float)nx.x, (float)nx.y);
                // the per-agent vectors of reset_world (nf:233-238, :446-459), by the agent's own lane: into the registers a span
                // carries, else into the state
                if (p.has_max_speed) mtime = dist2(nx, t.pos()[N + i]) / p.max_speed;   // (goal_match reset to arange BEFORE min_time)
                if (keep) {
                    c.x = nx; c.v = make_double2(0.0, 0.0); c.pd = 0.0; c.Dg = -1.0; c.Tr = -1.0; c.left = -1.0;
                    c.bits = fairnav_pack(0.0, 0, -1.0, 0); c.hits = 0;
                }
                else {   // (a span's inner step: the last step of the launch stores these from the registers)
                    p.agent_pos[g] = nx; p.agent_vel[g] = make_double2(0.0, 0.0); p.p_dist[g] = 0.0; p.status[g] = 0;
                    p.times_required[g] = -1.0; p.dists_to_goal[g] = -1.0; p.dist_left[g] = -1.0; p.goal_reached[g] = -1;
                    p.num_obst_coll[g] = 0; p.num_agent_coll[g] = 0; p.goal_occ[g] = 0.0; p.goal_history[g] = -1;
                    if (i == 0) p.cur_step[env] = 0;
                }
                if (p.has_max_speed) p.min_time[g] = mtime;   // (constant over the episode: not carried, re-read by the info planes)
            }
            __syncthreads();
            if (ended) {
                const double2 nx = t.pos()[i];
                for (int k = 0; k < L; ++k) t.D()[i * L + k] = dist2(nx, t.pos()[N + k]);
                double m = 1e300;
                for (int a = 0; a < N; ++a) m = fmin(m, dist2(t.pos()[a], t.pos()[N + i]));
                t.minprox()[i] = m;
            }
            __syncthreads();
            // the assignment of the new episode (nf:469): nothing in this step reads it (the walk, the observation and the rows do not
            // know the assignment; every step assigns afresh inside reward(agent 0)) -- it is state, and inside a span the last step
            // of the launch stores its own: skipped there, with its barrier
            // (round 6: straight into the state by the lanes that solve it -- no table entry, no barrier, no store by the env's own lanes
            // behind one; the envs it is for are marked in words()[2], the flags go back to "every env emits" at once)
            if (!keep && !FMARL_SKIP(p, 64)) {
                int *gm = p.goal_match + (size_t)env0 * N;
                if (N <= 4) fairnav_assign_tasks<4, THREADS>(p, lds, nenv, gm);
                else if (N <= 8) fairnav_assign_tasks<8, THREADS>(p, lds, nenv, gm);
                else if (N <= 16) fairnav_assign_tasks<16, THREADS>(p, lds, nenv, gm);
                else fairnav_assign_tasks<32, THREADS>(p, lds, nenv, gm);
            }
            if (in_range && i == 0) *t.flag() = 0;   // every env emits (read again behind the walk's barriers)
        } else if (in_range && i == 0) *t.flag() = 0;    // (no env of the workgroup ended)
        FMARL_TICK(9);   // in-kernel reset of the ended envs
    }

    // ---- the sequential part: occupancy / history walk in agent order
    ObsGoal og;
    og.goal = -1; og.second = 0; og.g_occ = og.g_hist = og.second_occ = 0.0;
    // Three barriers per agent (round 6; four before): the clear that graph_observation(a) asks for (nf:1278) is applied by the lane that
    // runs the NEXT event, just before it -- nobody looks at the occupancy in between -- and the last one behind the loop.
    for (int a = 0; a < (FMARL_SKIP(p, 128) ? 0 : N); ++a) {
        if (active && i == a) {
            if (t.words()[0] != N) {   // graph_observation(a - 1) found no free goal: the clear itself, once, and the marker back to "nobody"
                for (int k = 0; k < L; ++k) t.occ()[k] = 0.0;
                t.words()[0] = N;
            }
            og = obs_event(t.D() + i * L, t.minprox(), t.occ(), t.hist(), L, i, p.thr, p.min_obs_dist);
        }
        __syncthreads();
        // graph_observation(a): row of entity i on the snapshot (nf:1255-1283)
        bool far = false, free_empty = true;
        int cc = 0, best = -1;
        if (active) {
            const double *Drow = t.D() + i * L;
            double dmin = Drow[0], bd = 1e300;
            for (int k = 0; k < L; ++k) {
                const double d = Drow[k];
                if (d < dmin) { dmin = d; cc = k; }
                if (t.occ()[k] != 1.0) { free_empty = false; if (d < bd) { bd = d; best = k; } }
            }
            far = !(dmin < p.min_obs_dist);
            if (far && free_empty) atomicMin(&t.words()[0], i);   // first entity that triggers the clear
        }
        __syncthreads();
        if (active) {
            const int astar = t.words()[0];                        // N = nobody
            const bool cleared = free_empty && astar < i;         // an earlier entity already cleared the flags
            NavRow r;
            r.pad = 0;
            if (!far) { r.code = (int8_t)cc; r.occ = cleared ? 0.f : (float)t.occ()[cc]; r.hist = t.hist()[cc]; }
            else if (!free_empty) { r.code = (int8_t)best; r.occ = (float)t.occ()[best]; r.hist = t.hist()[best]; }
            else if (astar == i) { r.code = -1; r.occ = 0.f; r.hist = t.hist()[i]; }
            else { r.code = (int8_t)cc; r.occ = 0.f; r.hist = t.hist()[cc]; }   // after the clear every goal is free
            t.rows()[a * N + i] = r;
        }
        __syncthreads();   // (the rows above read the occupancy and the marker: the next event's lane changes both)
    }
    if (!FMARL_SKIP(p, 128)) {
        if (active && i == 0 && t.words()[0] != N) {   // the last graph_observation's clear
            for (int k = 0; k < L; ++k) t.occ()[k] = 0.0;
            t.words()[0] = N;
        }
        __syncthreads();
    }
    FMARL_TICK(5);   // walk

    if (active) {
        // position and velocity as the tables hold them (an env that was reset above: its new episode's; agentf / posf are the
        // float32 roundings the observation carries)
        const double2 xo = t.pos()[i];
        const double2 goal = og.goal >= 0 ? t.pos()[N + og.goal] : xo;
        const double2 sec = t.pos()[N + og.second];
        if (o.obs && emit) {   // nf:997-1000
            float *ob = o.obs + g * p.D;
            const float4 af = t.agentf()[i];
            const float2 pf = t.posf()[i];
            ob[0] = af.x; ob[1] = af.y; ob[2] = pf.x; ob[3] = pf.y;
            ob[4] = (float)(goal.x - xo.x); ob[5] = (float)(goal.y - xo.y); ob[6] = (float)og.g_occ; ob[7] = (float)og.g_hist;
            ob[8] = (float)(sec.x - xo.x); ob[9] = (float)(sec.y - xo.y); ob[10] = (float)og.second_occ;
        }
        if (keep) { c.occ = t.occ()[i]; c.bits = (c.bits & ~(0xffu << 1)) | ((uint32_t)((int)t.hist()[i] + 1) & 0xffu) << 1; }
        else if (STEP || emit) { p.goal_occ[g] = t.occ()[i]; p.goal_history[g] = t.hist()[i]; }
        if (o.graph_record && emit) {   // what a learner on another GPU needs to rebuild this env's node_obs (fmarl.h)
            uint32_t *r = o.graph_record + g * (size_t)(5 + 3 * N);
            const float4 af = t.agentf()[i];
            const float2 pf = t.posf()[i];
            r[0] = __float_as_uint(pf.x); r[1] = __float_as_uint(pf.y); r[2] = __float_as_uint(af.x); r[3] = __float_as_uint(af.y);
            r[4] = __float_as_uint(af.z);
            for (int e = 0; e < N; ++e) {
                const NavRow nr = t.rows()[i * N + e];
                r[5 + 3 * e] = (uint32_t)(int32_t)nr.code; r[6 + 3 * e] = __float_as_uint(nr.occ); r[7 + 3 * e] = __float_as_uint((float)nr.hist);
            }
        }
    }
    FMARL_TICK(7);   // obs, occupancy state, record
    // ---- emission (rows table, positions, velocities are final since the loop's last barrier)
    if (FMARL_SKIP(p, 32)) { FMARL_TICKS_END; return; }
    fairnav_emit_rows<STEP, THREADS>(p, o, lds, env0, nenv);   // (a step pass writes all rows: no env keeps its previous ones)
    FMARL_TICK(8);   // node rows
    emit_adj<true>(p, o, lds, env0, 0, nenv, threadIdx.x, THREADS);
    FMARL_TICK(10);   // adj
    FMARL_TICKS_END;
}

template <bool STEP, int THREADS, int NL>
__global__ __launch_bounds__(THREADS, THREADS == 192 ? 3 : 4) void fairnav_kernel(Params p, FmarlOutputs o, const int32_t *action_idx,
                                                                 const float *action_vec, int auto_reset) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // the shapes and table offsets re-read from the argument block where they are used (fmarl_dev.h span_params_reloaded) instead of
    // all being loaded at the top and spilled into vector lanes: 144 -> 30 spilled scalar registers, 0.0575 -> 0.0555 ms per launch
    // at 65 536 x 3 (-5 % with episodes ending at all phases, profiles/r4_notes.md)
    FairnavCarry c;
    fairnav_pass<STEP, THREADS, NL>(span_params_reloaded(), o, lds, action_idx, action_vec, auto_reset, c, 0);
}

// fmarl_step_span for nav_fairassign_fairrew_formation_graph: T steps of the workgroup's own envs in one launch, episode ends
// included (the step resets its ended envs itself).  Between the steps the agent's state stays in registers (FairnavCarry: only the
// first step loads it, only the last one stores it), the static entities and the envs' episode counters in the LDS tables.  Round 4 sent the state through global memory every step (HBM traffic 1.34 x algorithmic): at 256 threads per workgroup the
// carry did not fit 128 registers (181, or scratch); with the agent-less fourth wave gone (THREADS = 192) the budget is 168.
// What the span saves besides: what lies outside the waves' lifetimes -- a wave of the step launch lives 36-44 us of the launch's 60
// (dispatch, the write-back of the L2 at the kernel's end) and the next launch starts 6 us later (profiles/r4_ticks_fnav.txt).
// All arguments are one struct, re-read from the argument block inside the time loop (span_params_reloaded): held in scalar
// registers across a whole step -- shapes, eight output pointers, nine strides -- they spill twice as many registers.
struct FairnavSpanArgs { Params p; FmarlOutputs o; SpanStrides s; const int32_t *action_idx; const float *action_vec; int T, auto_reset; };
// Shapes with more than 192 agent lanes per workgroup (N >= 4) run 256 threads wide, and there the carry does not pay: at three
// workgroups per CU (168 registers, what the carry needs) the 10-agent shape took 0.775 ms per step against 0.634 for one launch per
// step at four (profiles/r6_fnav_spans_by_n.txt) -- its step is bound by the assignment and the sequential walk, which want the fourth
// workgroup's waves, not by the state's 100 bytes per agent.  So <256> sends the state through global memory between the steps (every
// field is read back by the lane -- or, the env's counters, by the workgroup -- that stored it, behind the barrier) at four per CU.
#ifndef FMARL_FNAV256_CARRY
#define FMARL_FNAV256_CARRY 0
#endif
#ifndef FMARL_FNAV256_BLOCKS
#define FMARL_FNAV256_BLOCKS (FMARL_FNAV256_CARRY ? 3 : 4)
#endif
template <int THREADS, int NL>
__global__ __launch_bounds__(THREADS, THREADS == 192 ? 3 : FMARL_FNAV256_BLOCKS) void fairnav_span_kernel(FairnavSpanArgs) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool kCarry = THREADS == 192 || FMARL_FNAV256_CARRY;
    FairnavCarry c = {};
    for (int t = 0;; ++t) {
        const FairnavSpanArgs &a = span_params_reloaded<FairnavSpanArgs>();
        if (t >= a.T) break;
        const FmarlOutputs ot = span_outputs(a.o, a.s, t);
        fairnav_pass<true, THREADS, NL>(a.p, ot, lds, a.action_idx ? a.action_idx + (size_t)t * a.s.actions : nullptr,
                                    a.action_vec ? a.action_vec + (size_t)t * a.s.actions : nullptr, a.auto_reset, c,
                                    kCarry ? ((t > 0 ? 1 : 0) | (t < a.T - 1 ? 2 : 0)) : 0);
        __syncthreads();   // the next step overwrites the LDS tables the emission read (and, without the carry, reads the state this one stored)
    }
}

// Learner-side reconstruction of node_obs / adj of nav_fairassign_fairrew_formation_graph envs from the gathered records:
// per agent and step [x, y, vx, vy, newly-stopped | (goal code, occupancy, history) x N] written by fairnav_kernel
// (FmarlOutputs.graph_record) + the once-per-episode record of the static entities (fmarl_rebuild.hip layout).  Same LDS
// tables, same emission code: bit-identical to the sender's.  n_envs is the caller's.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void fairnav_rebuild_kernel(Params p, FmarlOutputs o, const uint32_t *ep_rec,
                                                                  const uint32_t *step_rec, int n_envs) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, N = p.N;
    const int env0 = env_block(p) * p.epb;
    const int nenv = min(p.epb, n_envs - env0);
    const int el = tid / N, i = tid - el * N;
    const int LO = p.L + p.O, words = 2 * N + 2 * LO + 6 * p.W;   // episode_record_words (fmarl_rebuild.hip)
    if (el < nenv) {
        const FairNavLds t(p, lds, el);
        const uint32_t *r = step_rec + ((size_t)(env0 + el) * N + i) * (size_t)(5 + 3 * N);
        t.posf()[i] = make_float2(__uint_as_float(r[0]), __uint_as_float(r[1]));
        t.agentf()[i] = make_float4(__uint_as_float(r[2]), __uint_as_float(r[3]), __uint_as_float(r[4]), 0.f);
        for (int e = 0; e < N; ++e) {
            NavRow nr;
            nr.code = (int8_t)(int32_t)r[5 + 3 * e]; nr.occ = __uint_as_float(r[6 + 3 * e]); nr.hist = (int8_t)__uint_as_float(r[7 + 3 * e]); nr.pad = 0;
            t.rows()[i * N + e] = nr;
        }
        if (i == 0) *t.flag() = 0;
    }
    for (int k = tid; k < nenv * LO; k += THREADS) {
        const int e_l = k / LO, j = k - e_l * LO;
        const float *sp = (const float *)(ep_rec + (size_t)(env0 + e_l) * words) + 2 * (N + j);
        FairNavLds(p, lds, e_l).posf()[N + j] = make_float2(sp[0], sp[1]);
    }
    for (int k = tid; k < nenv * p.W; k += THREADS) {
        const int e_l = k / p.W, w = k - e_l * p.W;
        const FairNavLds t(p, lds, e_l);
        const uint32_t *q = ep_rec + (size_t)(env0 + e_l) * words + 2 * (N + LO) + 6 * w;
        const double axis = __longlong_as_double((long long)((unsigned long long)q[0] | ((unsigned long long)q[1] << 32)));
        const float *qf = (const float *)q;
        ((float4 *)(t.base + p.lds_wallf))[w] = make_float4(qf[2], (float)(axis + kWallWidth / 2), qf[3], (float)(axis - kWallWidth / 2));
        t.posf()[N + LO + w] = qf[4] == 0.f ? make_float2(0.f, (float)axis) : make_float2((float)axis, 0.f);
    }
    __syncthreads();
    fairnav_emit_rows<false, THREADS>(p, o, lds, env0, nenv);
    emit_adj(p, o, lds, env0, 0, nenv, threadIdx.x, THREADS);
}

}  // namespace fmarl
