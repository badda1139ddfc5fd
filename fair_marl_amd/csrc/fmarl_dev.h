// Device-side definitions shared by the gfx950 kernels of libfmarl (CDNA4 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fmarl.h"

// Phase ablation for tools/ablate.sh: compiled in only with -DFMARL_MEASURE (the shipped library has no such switch).
#ifdef FMARL_MEASURE
#define FMARL_SKIP(p, bit) (((p).ablate & (bit)) != 0)
// Phase clocks (tools/phase_ticks.py): every wave adds the core-clock cycles between two FMARL_TICK sites to its own row of a
// device table; the host averages the rows.  FMARL_TICKS_END writes the row.
#define FMARL_TICK_PHASES 16
#define FMARL_TICK_ROWS 65536
__device__ unsigned int g_fmarl_ticks[FMARL_TICK_ROWS][FMARL_TICK_PHASES];
// counters of the slot matchings (tools/phase_ticks.py cfg4 with a -DFMARL_MEASURE -DFMARL_HSTAT build: the atomics distort the
// clocks, so they have a switch of their own): [which][tasks run, tasks skipped, rows through the augmenting search, search iterations]
__device__ unsigned long long g_fmarl_hstat[2][4];
#define FMARL_TICKS_BEGIN unsigned int ticks_[FMARL_TICK_PHASES] = {}; unsigned long long tick_ = wall_clock64(); ticks_[15] = (unsigned int)tick_; tick_ = clock64();
#define FMARL_TICK(k) do { const unsigned long long now_ = clock64(); ticks_[k] += (unsigned int)(now_ - tick_); tick_ = now_; } while (0)
#define FMARL_TICKS_END do { ticks_[14] = (unsigned int)wall_clock64(); const unsigned int row_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); \
    if ((threadIdx.x & 63) == 0 && row_ < FMARL_TICK_ROWS) for (int k_ = 0; k_ < FMARL_TICK_PHASES; ++k_) g_fmarl_ticks[row_][k_] = ticks_[k_]; } while (0)
#else
#define FMARL_SKIP(p, bit) false
#define FMARL_TICKS_BEGIN
#define FMARL_TICK(k) do { } while (0)
#define FMARL_TICKS_END do { } while (0)
#endif

namespace fmarl {

// World constants: reference multiagent/core.py:153-161, :68, :38; environment.py:307.
constexpr double kDt = 0.1;
constexpr double kDamping = 0.25;
constexpr double kContactForce = 3e2;
constexpr double kContactMargin = 2e-2;
constexpr double kWallContactForce = 2.2e2;
constexpr double kWallContactMargin = 2.4e-2;
constexpr double kEntitySize = 0.05;
constexpr double kWallWidth = 0.1;
constexpr double kSensitivity = 5.0;
constexpr int kThreads = 256;       // workgroup size of the step / emit kernels (4 waves)
constexpr int kMaxTries = 10000;    // bound on the reference's unbounded rejection loops
constexpr int kEgoWidth = 5;         // LDS ego row: vx, vy, x, y, 0
constexpr int kStageRows = 64;      // rows per wave and window of the generic emission (one row per lane)
constexpr int kStepWavesPerSimd = 4; // register budget of step_kernel (128 VGPRs); measured best of 3..6 on MI355X

// floor(q / d) for q * d < 2^40, q < 2^24 (block-local flat indices): one 64-bit multiply.
struct FastDiv {
    uint64_t m;
    uint32_t d;
    __host__ void set(uint32_t div) { d = div; m = ((1ull << 40) + div - 1) / div; }
    __device__ __forceinline__ uint32_t div(uint32_t q) const { return (uint32_t)(((uint64_t)q * m) >> 40); }
};

// A wave-uniform value made opaque to the compiler.  Kernel arguments are "free to reload": when scalar registers run short
// inside a loop the register allocator re-reads them from the kernel-argument block at every use (this build has no machine
// LICM to hoist them back), and each s_load's s_waitcnt lgkmcnt(0) also waits for the LDS operations in flight.  A pinned
// copy has to stay in an SGPR (or be spilled to a VGPR lane: one VALU move, no wait).
__device__ __forceinline__ uint32_t pin_sgpr(uint32_t v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ int pin_sgpr(int v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ uint64_t pin_sgpr(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return ((uint64_t)hi << 32) | lo;
}

// Kernel parameter block (passed by value).  State pointers follow include/fmarl.h FMARL_F_*.
struct Params {
    int n_envs, N, L, O, W, E, D, F;
    int episode_length, has_max_speed, env_offset, scenario;
    int epb;                 // environments per workgroup = kThreads / N
    int order;               // workgroup b works on env block (b * order) mod gridDim.x (env_block below): 1 = dispatch order
    int epw;                 // formation scenario: environments per wave (every env inside one wave), epb = 4 epw
    int feat_global;   // FMARL_FLAG_GLOBAL_FEATURES: node rows are [vel, pos, goal, type] without the ego part
    int lds_pos, lds_agentf, lds_ego, lds_stat, lds_wall, lds_flag, lds_env_bytes;  // per-env LDS byte offsets
    int lds_posf, has_posf;  // navigation_graph: f32 copy of the entity positions (adj is computed from it)
    int lds_wallf, lds_constf;   // f32 wall corner words (4 per wall; not for the formation scenario) and, navigation_graph, the constants 0, 1, 2, 3
    int has_wallf, lds_cnt;      // wall corner table present; byte offset of the env's policy-edge counter (fused processAdj count)
    double world_size, max_speed, collision_rew, goal_rew, thr, fair_rew, zeroshift;
    float edge_thr;          // (float)max_edge_dist: the policy-edge threshold of the fused processAdj count
    uint64_t seed;
    FastDiv dNEF, dEF, dF, dEE, dE, dNE, dC4, dNC4, dEE4, dE4;
    FastDiv dP, dN;          // small batches (step_body SMALL): contact partners per agent (N + O + W), agents per env
    int ablate;              // -DFMARL_MEASURE builds only (tools/ablate.sh): bit mask of phases to skip
    int vec_node, vec_adj;   // 16-byte emission paths usable (E*F % 4 == 0 / E % 4 == 0)
    int scan_stats;          // N is a power of two <= 64: the agent-loop statistics run as wave scans (seg_mixed_stats)
    int lds_stage, stage_wave_bytes;   // generic shapes: per-wave LDS window the rows go through (offset from the LDS base, bytes per wave)
    int stat_stride;   // statistics block of env el: LDS base + lds_stat + el * stat_stride (lds_env_bytes, or 5 N doubles when the blocks share the emission windows' region)
    // formation scenario: extra per-env LDS tables (byte offsets) and state
    int f_slot_new, f_slot_old, f_g, f_masks, f_theta, f_words, f_slotf;
    int lds2_bytes, f_rows;   // second LDS region, lds_stage + el * lds2_bytes: tables dead once the emission starts (formation: lds_stat,
                              // f_slot_old, f_theta, f_words; fairnav: lds_stat, n_D, n_minprox, n_occ, n_match, n_words) / formation: rows per window
    double2 *slot_pos;
    double *slot_occ, *slot_delta, *formation_done, *match_dual;
    const double2 *rot_table;   // (cos, sin) of i * 2 pi / N, one table for all envs
    // fairnav scenario: extra per-env LDS tables (byte offsets), knob and state
    int n_D, n_minprox, n_occ, n_match, n_rows, n_words;
    int n_pre;   // Philox blocks of an env's new episode drawn ahead by all lanes in the in-kernel reset (they fit the env's part of the second region below n_words)
    double min_obs_dist;
    double *goal_occ;
    int8_t *goal_history, *goal_reached, *status;   // integer-valued in the reference (-1 / index, -1 / index, bool)
    // state
    double2 *agent_pos, *agent_vel, *landmark_pos, *obstacle_pos;
    double *p_dist, *wall_axis, *wall_e0, *wall_e1, *wall_length;
    double *dists_to_goal, *times_required, *dist_left, *min_time;
    int *wall_orient, *goal_match, *num_obst_coll, *num_agent_coll, *cur_step, *episode, *reset_flag;
    // staged next episode (FMARL_FLAG_ASYNC_RESET)
    double2 *st_agent_pos, *st_landmark_pos, *st_obstacle_pos;
    double *st_wall_axis;
    int *st_wall_orient, *st_goal_match, *stage_valid, *stage_need;
    int *place_fails, *st_place_fails;   // placements accepted after kMaxTries colliding draws (live / staged episode)
};

// Which block of p.epb consecutive envs this workgroup works on.  Workgroups are dispatched in index order, so with the identity
// mapping the ~1 000 workgroups resident at any time write one compact, moving window of the output arrays; HBM takes a store
// stream 10-20 % faster when the concurrently written regions are scattered over the whole buffer instead (pure 16-byte store
// streams on MI355X: 5.9-6.8 TB/s in dispatch order, 7.1 TB/s scattered, for every chunk size from 64 KB to 4 MB:
// tools/store_ceiling.py, profiles/r4_store_ceiling.md).  `order` is coprime with the grid (a permutation; the host picks the
// odd number nearest 0.618 x grid: scatter_order in libfmarl.hip), results never depend on it.
__device__ __forceinline__ int env_block(const Params &p) {
    return p.order > 1 ? (int)(((uint64_t)blockIdx.x * (uint32_t)p.order) % gridDim.x) : (int)blockIdx.x;
}

// fmarl_step_span: per-step strides (elements) of the outputs and of the action tape.
struct SpanStrides { long long obs, node_obs, adj, reward, done, info, edge_nnz, graph_record, actions; };
__device__ __forceinline__ FmarlOutputs span_outputs(const FmarlOutputs &o, const SpanStrides &s, int t) {
    FmarlOutputs ot = o;
    if (ot.obs) ot.obs += (size_t)t * s.obs;
    if (ot.node_obs) ot.node_obs += (size_t)t * s.node_obs;
    if (ot.adj) ot.adj += (size_t)t * s.adj;
    if (ot.reward) ot.reward += (size_t)t * s.reward;
    if (ot.done) ot.done += (size_t)t * s.done;
    if (ot.info) ot.info += (size_t)t * s.info;
    if (ot.edge_nnz) ot.edge_nnz += (size_t)t * s.edge_nnz;
    if (ot.graph_record) ot.graph_record += (size_t)t * s.graph_record;
    return ot;
}
// The shapes and table offsets of `p`, made opaque: what a step derives from them once per launch (the emission's source-offset
// tables, uses of the fast-division constants) must not be hoisted out of a span kernel's time loop and stay live across a whole
// step -- the kernel would spill.
__device__ __forceinline__ Params span_params(const Params &p) {
    Params q = p;
    const int z = pin_sgpr(0);   // an opaque zero: (invariant + z) is loop-variant to the compiler, yet can be re-formed at every use
#define FMARL_PIN(f) q.f = p.f + z;
    FMARL_PIN(N) FMARL_PIN(E) FMARL_PIN(F) FMARL_PIN(L) FMARL_PIN(O) FMARL_PIN(W) FMARL_PIN(D) FMARL_PIN(epb) FMARL_PIN(epw) FMARL_PIN(n_envs)
    FMARL_PIN(lds_env_bytes) FMARL_PIN(lds_pos) FMARL_PIN(lds_agentf) FMARL_PIN(lds_ego) FMARL_PIN(lds_stat) FMARL_PIN(lds_wall)
    FMARL_PIN(lds_flag) FMARL_PIN(lds_posf) FMARL_PIN(lds_wallf) FMARL_PIN(lds_constf) FMARL_PIN(lds_cnt) FMARL_PIN(lds_stage)
    FMARL_PIN(stage_wave_bytes) FMARL_PIN(stat_stride) FMARL_PIN(vec_node) FMARL_PIN(vec_adj) FMARL_PIN(scan_stats)
    FMARL_PIN(feat_global) FMARL_PIN(has_wallf) FMARL_PIN(has_posf) FMARL_PIN(lds2_bytes) FMARL_PIN(f_rows)
    FMARL_PIN(f_slot_new) FMARL_PIN(f_slot_old) FMARL_PIN(f_g) FMARL_PIN(f_masks) FMARL_PIN(f_theta) FMARL_PIN(f_words)
    FMARL_PIN(n_D) FMARL_PIN(n_minprox) FMARL_PIN(n_occ) FMARL_PIN(n_match) FMARL_PIN(n_rows) FMARL_PIN(n_words)
    FMARL_PIN(dNEF.m) FMARL_PIN(dEF.m) FMARL_PIN(dF.m) FMARL_PIN(dEE.m) FMARL_PIN(dE.m) FMARL_PIN(dNE.m) FMARL_PIN(dC4.m) FMARL_PIN(dNC4.m)
    FMARL_PIN(dEE4.m) FMARL_PIN(dE4.m)
#undef FMARL_PIN
    return q;
}
// The same purpose by another route (formation_span_kernel): the kernel's argument block re-read through a pointer the compiler
// cannot see through -- the fields are then loaded (scalar loads from the constant segment) where a step uses them instead
// of all being loaded before the time loop and kept in scalar registers across it (span_params keeps ~55 of them live: the
// kernel spilled 194 scalar registers into vector lanes, each access a VALU instruction).  `Params` must be the first argument.
template <typename Args = Params>   // (Args: a kernel's ONLY argument, or the type of its first one)
__device__ __forceinline__ const Args &span_params_reloaded() {
    typedef const Args __attribute__((address_space(4))) *ConstArgs;
    uint64_t a = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
    a = pin_sgpr(a);
    return *(const Args *)(ConstArgs)a;
}
// between two steps of a span: a workgroup re-reads from global memory what it wrote itself
__device__ __forceinline__ void span_step_done() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
}

// Ordering point between phases of ONE wave that talk through LDS (kernels whose envs each live inside one wave): LDS
// executes a wave's instructions in order, the fence keeps the compiler from moving accesses across.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------- Philox4x32-10 (oracle/philox.py)
constexpr uint32_t kPhiloxM0 = 0xD2511F53u, kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kPhiloxW0 = 0x9E3779B9u, kPhiloxW1 = 0xBB67AE85u;
constexpr uint32_t kPhiloxTag = 0x464D4152u;

// block `idx` of the stream (seed; idx, env, episode) -> two doubles in [0, 1) built like NumPy's random_double
__device__ __forceinline__ void philox_block(uint64_t seed, uint32_t idx, uint32_t env, uint32_t episode, double &u0, double &u1) {
    uint32_t c0 = idx, c1 = env, c2 = episode, c3 = kPhiloxTag, a = (uint32_t)seed, b = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r) { a += kPhiloxW0; b += kPhiloxW1; }
        uint64_t p0 = (uint64_t)kPhiloxM0 * c0, p1 = (uint64_t)kPhiloxM1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    }
    const double s = 1.0 / 9007199254740992.0;
    u0 = (double)((((uint64_t)c1 << 32) | c0) >> 11) * s;
    u1 = (double)((((uint64_t)c3 << 32) | c2) >> 11) * s;
}

struct PhiloxStream {
    uint64_t seed;
    uint32_t env, episode, idx, n_pre;
    const double *pre;   // blocks 0 .. n_pre - 1 of this stream, drawn ahead by other lanes (u0, u1 pairs; the in-kernel reset of fairnav_pass)
    __device__ PhiloxStream(uint64_t seed_, uint32_t env_, uint32_t episode_, uint32_t first = 0, const double *pre_ = nullptr, uint32_t n_pre_ = 0)
        : seed(seed_), env(env_), episode(episode_), idx(first), n_pre(n_pre_), pre(pre_) {}
    // one 128-bit block -> two doubles in [0, 1)
    __device__ void next(double &u0, double &u1) {
        if (idx < n_pre) { u0 = pre[2 * idx]; u1 = pre[2 * idx + 1]; ++idx; return; }
        philox_block(seed, idx++, env, episode, u0, u1);
    }
    __device__ double2 uniform_pair(double lo, double hi) {
        double a, b; next(a, b);
        return make_double2(lo + (hi - lo) * a, lo + (hi - lo) * b);
    }
    __device__ double uniform(double lo, double hi) { double a, b; next(a, b); return lo + (hi - lo) * a; }
    __device__ int choice_hv() { double a, b; next(a, b); return a < 0.5 ? 0 : 1; }  // 0 = 'H'
};

// log(u) for u in [1, 2], absolute error < 1e-16: u -> m in [1/sqrt2, sqrt2], s = (m-1)/(m+1),
// log m = 2 s (1 + s^2/3 + ... + s^20/21) (|s| <= 0.172, truncation 6e-19).  ~27 f64 ops where the
// library log1p spends 113; absolute (not relative) accuracy is what the force sum needs.
// 1 / x to an ulp or so for normal x: v_rcp_f64 + two Newton steps (a correctly rounded f64 division is 12 instructions)
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

__device__ __forceinline__ double log_1to2(double u) {
    const bool big = u > 1.4142135623730951;
    const double m = big ? u * 0.5 : u;
    const double s = (m - 1.0) * rcp_nr(m + 1.0), w = s * s;
    double q = 1.0 / 21.0;
    q = fma(q, w, 1.0 / 19.0); q = fma(q, w, 1.0 / 17.0); q = fma(q, w, 1.0 / 15.0);
    q = fma(q, w, 1.0 / 13.0); q = fma(q, w, 1.0 / 11.0); q = fma(q, w, 1.0 / 9.0);
    q = fma(q, w, 1.0 / 7.0);  q = fma(q, w, 1.0 / 5.0);  q = fma(q, w, 1.0 / 3.0);
    q = fma(q, w, 1.0);
    const double r = 2.0 * s * q;
    return big ? r + 0.6931471805599453 : r;
}

// For values that only leave as float32 OUTPUTS (the fairness column of obs, reward, info planes): a quotient by a
// reciprocal (an ulp or two off the correctly rounded division, 8 instructions instead of ~35) and tanh through one exp
// (absolute error ~1e-16; the library's tanh is ~170 instructions).  Never used on state.
__device__ __forceinline__ double ratio_out(double num, double den) { return num * rcp_nr(den); }
__device__ __forceinline__ double tanh_out(double z) {
    const double t = exp(-2.0 * fabs(z));   // (0, 1]
    return copysign((1.0 - t) * rcp_nr(1.0 + t), z);
}

// np.logaddexp(0, z) * k -- reference core.py:391 / :439 (softplus penetration).
// logaddexp(0, z) = max(z, 0) + log(1 + exp(-|z|)); 1 + t is rounded once (abs error 1.1e-16).
__device__ __forceinline__ double softplus_pen(double z, double k) {
    const double t = exp(-fabs(z));
    return (fmax(z, 0.0) + log_1to2(1.0 + t)) * k;
}

// sqrt(x) for x = 0 or x in the normal range (squared distances inside a 2 x 2 world): the library's own
// Goldschmidt iteration on v_rsq_f64 (same bits) without its rescaling of tiny / huge arguments and its class test,
// 13 instructions instead of 25 -- this sits inside the contact pairs and the matchings' inner loops.
__device__ __forceinline__ double sqrt_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    g = fma(fma(-g, g, x), h, g);
    return x == 0.0 ? 0.0 : g;
}

// sqrt(x) and 1 / sqrt(x) for x > 0 in the normal range (the contact pairs need both: d and the unit vector)
__device__ __forceinline__ double sqrt_inv_pos(double x, double &inv) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    g = fma(fma(-g, g, x), h, g);
    const double i2 = h + h;                 // 1 / sqrt(x) after one refinement; one Newton step against g
    inv = fma(fma(-g, i2, 1.0), i2, i2);
    return g;
}

__device__ __forceinline__ double dist2(double2 a, double2 b) {
    double dx = a.x - b.x, dy = a.y - b.y;
    return sqrt_pos(dx * dx + dy * dy);
}

// |(dx, dy)| for the f32 outputs: one v_sqrt_f32 (1 ulp; sqrtf's correctly rounded sequence is 17 instructions,
// and the adj emission evaluates 5 184 of these per env and step at E = 72)
__device__ __forceinline__ float dist_f32(float dx, float dy) { return __builtin_amdgcn_sqrtf(fmaf(dy, dy, dx * dx)); }

// sqrt(|a-b|^2) < c, decided on the squared distance unless it is within a few ulp of c^2
// (navigation_graph.py:659, :695, :705 compare np.linalg.norm(delta) < dist_min)
__device__ __forceinline__ bool closer_than(double2 a, double2 b, double c) {
    const double dx = a.x - b.x, dy = a.y - b.y, s = dx * dx + dy * dy, c2 = c * c;
    if (s < c2 * (1.0 - 1e-15)) return true;
    if (s > c2 * (1.0 + 1e-15)) return false;
    return sqrt(s) < c;
}

// navigation_graph.py:671-683: wall box test with the 1.05 factors
__device__ __forceinline__ bool wall_box_hit(double2 p, double axis, double e0, double e1, int orient) {
    const double s = kEntitySize;
    double pperp = orient == 0 ? p.y : p.x, ppar = orient == 0 ? p.x : p.y;
    return (1.05 * (axis - s / 2) <= pperp) && (pperp <= 1.05 * (axis + s / 2)) &&
           (1.05 * (e0 - s / 2) <= ppar) && (ppar <= 1.05 * (e1 + s / 2));
}

// Lane exchange inside a 16-lane DPP row (no LDS crossbar, a few cycles instead of a ds_bpermute round
// trip): quad_perm xor 1, quad_perm xor 2, row_half_mirror, row_mirror.  For a symmetric reduction
// (min / argmin) the mirrors do the job of xor 4 / xor 8: after the quad steps every quad is uniform.
// Permutations (quad_perm, row_mirror, row_half_mirror) give every lane a source: mov_dpp, whose destination needs no
// seed.  Shifts and broadcasts leave some lanes without one: those keep their own value (update_dpp with old = v), which
// costs a v_mov to seed the destination -- the scans rely on it.
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) {
    if constexpr (CTRL <= 0xFF || CTRL == 0x140 || CTRL == 0x141) return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
    else return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
// min / max of a double over a group of G lanes, result in every lane
template <int G, bool MAX> __device__ __forceinline__ double group_extreme(double v) {
    // one v_min_f64 / v_max_f64 per step (asm: the builtin form canonicalises both operands first; no NaNs here)
#define FMARL_EXT_STEP(o) { const double o_ = (o); if (MAX) asm("v_max_f64 %0, %1, %2" : "=v"(v) : "v"(v), "v"(o_)); \
                            else asm("v_min_f64 %0, %1, %2" : "=v"(v) : "v"(v), "v"(o_)); }
    FMARL_EXT_STEP(dpp_f64<0xB1>(v))                 // quad_perm [1,0,3,2]
    FMARL_EXT_STEP(dpp_f64<0x4E>(v))                 // quad_perm [2,3,0,1]
    if (G >= 8) FMARL_EXT_STEP(dpp_f64<0x141>(v))    // row_half_mirror
    if (G >= 16) FMARL_EXT_STEP(dpp_f64<0x140>(v))   // row_mirror
#pragma unroll
    for (int off = 16; off < G; off <<= 1) FMARL_EXT_STEP(__shfl_xor(v, off, G))
#undef FMARL_EXT_STEP
    return v;
}
// lowest (LAST = false) or highest lane index inside the group whose predicate holds; G if none
template <int G, bool LAST> __device__ __forceinline__ int group_pick_lane(bool pred) {
    const unsigned long long m = __ballot(pred);
    const int base = (threadIdx.x & 63) & ~(G - 1);
    const unsigned long long g = G == 64 ? m : (m >> base) & ((1ull << (G & 63)) - 1);
    if (g == 0) return G;
    return LAST ? 63 - __builtin_clzll(g) : __builtin_ctzll(g);
}
// (value, lane) argmin over a group: smallest value, ties to the lowest lane; lanes with valid = false do not
// take part.  One value reduction + one ballot instead of reducing (value, index) pairs.
template <int G> __device__ __forceinline__ void group_argmin(double v, bool valid, double &best, int &lane_out) {
    best = group_extreme<G, false>(valid ? v : __builtin_huge_val());
    lane_out = group_pick_lane<G, false>(valid & (v == best));
}
// (value, lane) argmax: largest value, ties to the highest lane
template <int G> __device__ __forceinline__ void group_argmax(double v, bool valid, double &best, int &lane_out) {
    best = group_extreme<G, true>(valid ? v : -__builtin_huge_val());
    lane_out = group_pick_lane<G, true>(valid & (v == best));
}
// bitwise OR over a group of G lanes, result in every lane
template <int G> __device__ __forceinline__ uint32_t group_or32(uint32_t v) {
    v |= (uint32_t)dpp_i32<0xB1>((int)v);
    v |= (uint32_t)dpp_i32<0x4E>((int)v);
    if (G >= 8) v |= (uint32_t)dpp_i32<0x141>((int)v);
    if (G >= 16) v |= (uint32_t)dpp_i32<0x140>((int)v);
#pragma unroll
    for (int off = 16; off < G; off <<= 1) v |= (uint32_t)__shfl_xor((int)v, off, G);
    return v;
}

// ---------------------------------------------------------------- run statistics by wave scans
// (mean, M2 = sum of squared deviations from the mean) of runs of consecutive lanes, joined pairwise
// (Chan, Golub, LeVeque): no cancellation, and a run of equal values keeps M2 = 0 exactly -- np.std of a constant
// vector is exactly 0 and the fairness scalar divides by std + 1e-4 (navigation_graph.py:766, :769).
// Segments are the N lanes of one env, N a power of two (N | 64), li = lane index inside the segment.
__device__ __forceinline__ double rcp_small(double n) { return rcp_nr(n); }   // 1 / n for the small integer counts
// run A (ca values) joined with run B (cb values), both counts >= 1
__device__ __forceinline__ void run_join(double ma, double qa, double ca, double mb, double qb, double cb, double &m, double &q) {
    const double w = cb * rcp_small(ca + cb), d = mb - ma;
    m = ma + d * w;
    q = qa + qb + d * d * (ca * w);
}
// One scan step: lanes with take join the run (pm, pq) of cp values fetched from the partner lane with their own of co values.
#define FMARL_RUN_STEP(pm_, pq_, take_, cp_, co_) {                                                                  \
        const double pm = (pm_), pq = (pq_);                                                                         \
        const bool take = (take_);                                                                                   \
        double jm, jq;                                                                                               \
        run_join(pm, pq, take ? (double)(cp_) : 1.0, m, q, take ? (double)(co_) : 1.0, jm, jq);                      \
        m = take ? jm : m; q = take ? jq : q; }
// inclusive prefix: on return (m, q) of lane li describe the values of lanes [0, li] of its segment
__device__ __forceinline__ void seg_prefix_runs(int N, int li, double v, double &m, double &q) {
    m = v; q = 0.0;
    const int s = li & 15;   // DPP rows are 16 lanes: row_shr inside them, row_bcast across
    if (N > 1) FMARL_RUN_STEP(dpp_f64<0x111>(m), dpp_f64<0x111>(q), s >= 1, 1, 1)
    if (N > 2) FMARL_RUN_STEP(dpp_f64<0x112>(m), dpp_f64<0x112>(q), s >= 2, min(s - 1, 2), min(s + 1, 2))
    if (N > 4) FMARL_RUN_STEP(dpp_f64<0x114>(m), dpp_f64<0x114>(q), s >= 4, min(s - 3, 4), min(s + 1, 4))
    if (N > 8) FMARL_RUN_STEP(dpp_f64<0x118>(m), dpp_f64<0x118>(q), s >= 8, min(s - 7, 8), min(s + 1, 8))
    if (N > 16) FMARL_RUN_STEP(dpp_f64<0x142>(m), dpp_f64<0x142>(q), (li & 16) != 0, 16, s + 1)          // row_bcast:15
    if (N > 32) FMARL_RUN_STEP(dpp_f64<0x143>(m), dpp_f64<0x143>(q), (li & 32) != 0, 32, (li & 31) + 1)  // row_bcast:31
}
// inclusive suffix: on return (m, q) of lane li describe the values of lanes [li, N - 1] of its segment
__device__ __forceinline__ void seg_suffix_runs(int N, int li, double v, double &m, double &q) {
    m = v; q = 0.0;
    const int R = N < 16 ? N : 16, t = R - 1 - (li & (R - 1));   // lanes of the segment after this one inside its DPP row
    if (N > 1) FMARL_RUN_STEP(dpp_f64<0x101>(m), dpp_f64<0x101>(q), t >= 1, 1, 1)
    if (N > 2) FMARL_RUN_STEP(dpp_f64<0x102>(m), dpp_f64<0x102>(q), t >= 2, min(t - 1, 2), min(t + 1, 2))
    if (N > 4) FMARL_RUN_STEP(dpp_f64<0x104>(m), dpp_f64<0x104>(q), t >= 4, min(t - 3, 4), min(t + 1, 4))
    if (N > 8) FMARL_RUN_STEP(dpp_f64<0x108>(m), dpp_f64<0x108>(q), t >= 8, min(t - 7, 8), min(t + 1, 8))
    const int lane = threadIdx.x & 63;
    if (N > 16) {   // first lane of the next row holds that row's total
        const int src = (lane & ~31) | 16;
        FMARL_RUN_STEP(__shfl(m, src, 64), __shfl(q, src, 64), (li & 16) == 0, 16, t + 1)
    }
    if (N > 32) FMARL_RUN_STEP(__shfl(m, 32, 64), __shfl(q, 32, 64), li < 32, 32, 32 - (li & 31))
}
#undef FMARL_RUN_STEP
// all lanes of the segment: mean and M2 of the segment's N values (butterfly of equal-sized runs, symmetric in the
// two partners, so every lane ends with the same bits)
__device__ __forceinline__ void seg_all_runs(int N, double v, double &m, double &q) {
    m = v; q = 0.0;
#define FMARL_ALL_STEP(pm_, pq_, half_) { const double pm = (pm_), pq = (pq_), d = m - pm; \
        m = 0.5 * (m + pm); q = (q + pq) + d * d * (half_); }
    if (N > 1) FMARL_ALL_STEP(dpp_f64<0xB1>(m), dpp_f64<0xB1>(q), 0.5)      // quad_perm [1,0,3,2]
    if (N > 2) FMARL_ALL_STEP(dpp_f64<0x4E>(m), dpp_f64<0x4E>(q), 1.0)      // quad_perm [2,3,0,1]
    if (N > 4) FMARL_ALL_STEP(dpp_f64<0x141>(m), dpp_f64<0x141>(q), 2.0)    // row_half_mirror (quads are uniform by now)
    if (N > 8) FMARL_ALL_STEP(dpp_f64<0x140>(m), dpp_f64<0x140>(q), 4.0)    // row_mirror
    if (N > 16) FMARL_ALL_STEP(__shfl_xor(m, 16, 64), __shfl_xor(q, 16, 64), 8.0)
    if (N > 32) FMARL_ALL_STEP(__shfl_xor(m, 32, 64), __shfl_xor(q, 32, 64), 16.0)
#undef FMARL_ALL_STEP
}
// Statistics of the reference's sequential agent loop for lane li (navigation_graph.py:617-618, :769, :854):
// `info`  = mean / population std of [fresh_0 .. fresh_li, stale_li+1 .. stale_N-1]   (after info_callback(li))
// `before` = the same with fresh_li not yet in (what observation(li) / reward(li) see): info of lane li - 1, all stale for li = 0.
// Every lane of the wave must call this (no divergence around it).
__device__ __forceinline__ void seg_mixed_stats(int N, int li, double fresh, double stale, double &info_m, double &info_sd,
                                                double &before_m, double &before_sd) {
    double pm, pq, sm, sq;
    seg_prefix_runs(N, li, fresh, pm, pq);
    seg_suffix_runs(N, li, stale, sm, sq);
    const double nm = dpp_f64<0x130>(sm), nq = dpp_f64<0x130>(sq);   // wave_shl:1 -> suffix run starting at li + 1
    const bool last = li == N - 1;
    double jm, jq;
    run_join(pm, pq, (double)(li + 1), nm, nq, last ? 1.0 : (double)(N - 1 - li), jm, jq);
    info_m = last ? pm : jm;
    info_sd = sqrt((last ? pq : jq) / N);
    const double bm = dpp_f64<0x138>(info_m), bs = dpp_f64<0x138>(info_sd);   // wave_shr:1
    before_m = li == 0 ? sm : bm;
    before_sd = li == 0 ? sqrt(sq / N) : bs;
}

}  // namespace fmarl
