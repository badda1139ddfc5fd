// fair_graph_formation (BASELINE config 4) on gfx950: step + reset observation.
//
// Restates reference multiagent/custom_scenarios/fair_graph_formation.py ("ff:<line>") behind the same
// World.step physics (fmarl_step.hip world_step_agent):
//   ff:622-700 reward      slots on a circle about landmark 0 anchored at the smallest agent angle,
//                          agent x slot distances, occupancy, Hungarian matching (scipy at ff:615-618)
//   ff:703-741 observation three-way goal branch, concat(v, x, goal - x) + flag (list + ndarray quirk)
//   ff:810-971 graph_observation  12 features, agent rows run the same branch with the EGO's index
//   ff:441-501 info_callback      ring test, frozen dists_to_goal, statistics
// The reference walks the agents sequentially (environment.py:832-864) and the scenario mutates the
// slot occupancy inside observation / graph_observation; reward(agent 0) replaces slots and occupancy
// between observation(0) and observation(1).  Positions are fixed during that walk, so everything
// geometric is computed in parallel (one thread per agent) and only the occupancy bit-mask is walked
// sequentially by one lane per env (N (N + 1) branch events on a 32-bit mask).
#pragma once
#include "fmarl_dev.h"
#include "fmarl_kernels.h"
#include "fmarl_step.hip"

namespace fmarl {

constexpr double kTargetRadius = 0.5;   // ff:105

// Lane-to-lane reads inside a wave (ds_bpermute: `addr` = 4 x source lane).  A disabled source lane reads as 0: callers only
// read lanes of their own env's segment, which share their control flow.
__device__ __forceinline__ int bperm_i32(int addr, int v) { return __builtin_amdgcn_ds_bpermute(addr, v); }
__device__ __forceinline__ double bperm_f64(int addr, double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)b), hi = __builtin_amdgcn_ds_bpermute(addr, (int)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo));
}

// Min-sum assignment (Kuhn-Munkres with potentials, shortest augmenting paths) of ONE env by the env's own N agent lanes: lane
// `li` of the segment [base, base + N) of the wave owns column li (v, minv, way, matched row) and row li (u, in-tree flag).
// Every env of the wave runs its matchings at the same time (64 / N envs: 60 of 64 lanes at ten agents).  The cross-lane
// traffic: the argmin as a rotation all-reduce, a few single-lane reads (ds_bpermute).  Costs are |x_row - P_col| computed on the
// fly (x: agent positions in LDS, Pc: the lane's own slot).  The optimum is unique for generic real costs, so it equals SciPy's
// linear_sum_assignment (ff:615-618) whatever the start.
// Warm start: `v` = column potential left by an earlier matching on nearly the same costs (any values are feasible once u = the
// row minima of c - v: `u` / `mine` = reduced row minimum and its column, from the kernel's agent x slot distance pass);
// every row claims the column of its minimum, the lowest row wins a contested one (`claim`: an LDS table of N ints per
// matching), only the rows left without a column go through the augmenting search.
struct SegLanes {
    int N, li, base, self4, rot[7];
    uint32_t full;
    __device__ __forceinline__ SegLanes(int N_, int li_, int base_) : N(N_), li(li_), base(base_) {
        self4 = (base + li) << 2;
        // partners of the rotation all-reduce, radix 4: round r reads the lanes li + k 4^r (mod N), k = 1, 2, 3 -- three
        // independent ds_bpermute round trips in flight at once, then the window every lane has seen is four times as wide.
        // (Round 4 doubled the window per round: four DEPENDENT round trips at ten agents where this takes two.)
        constexpr int off[7] = {1, 2, 3, 4, 8, 12, 16};
#pragma unroll
        for (int s = 0; s < 7; ++s) {   // (N <= 32; fixed trip count: rot[] stays in registers)
            int q = li + off[s];
            q = q >= N ? q - N : q;
            rot[s] = (base + (off[s] < N ? q : li)) << 2;
        }
        full = N >= 32 ? ~0u : ((1u << N) - 1);
    }
    __device__ __forceinline__ int at(int k) const { return (base + k) << 2; }   // bpermute address of segment lane k
    template <bool MAX> static __device__ __forceinline__ double pick(double a, double b) {
        // one v_min_f64 / v_max_f64 (asm: the builtin form canonicalises both operands first; no NaNs here)
        if (MAX) asm("v_max_f64 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
        else asm("v_min_f64 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
        return a;
    }
    // min / max over the segment, in every lane of it (every lane ends with the extreme of a cyclic window >= N)
    template <bool MAX> __device__ __forceinline__ double extreme(double v) const {
        if (N > 1) {   // (all conditions uniform)
            const double o1 = bperm_f64(rot[0], v), o2 = N > 2 ? bperm_f64(rot[1], v) : v, o3 = N > 3 ? bperm_f64(rot[2], v) : v;
            v = pick<MAX>(pick<MAX>(v, o1), pick<MAX>(o2, o3));
        }
        if (N > 4) {
            const double o1 = bperm_f64(rot[3], v), o2 = N > 8 ? bperm_f64(rot[4], v) : v, o3 = N > 12 ? bperm_f64(rot[5], v) : v;
            v = pick<MAX>(pick<MAX>(v, o1), pick<MAX>(o2, o3));
        }
        if (N > 16) v = pick<MAX>(v, bperm_f64(rot[6], v));
        return v;
    }
    __device__ __forceinline__ uint32_t bits(bool pred) const {   // the segment's lanes whose predicate holds, bit k = lane k
        return (uint32_t)(__ballot(pred) >> base) & full;
    }
};

// The matchings of one step, as ONE loop: against the current slots (A: slot PA, row start uA / mineA -> ansA[row] = col, returns
// its final column potential shifted so that the maximum over the columns is 0: the next step's warm start) and, when `needB`,
// against the previous slots (B -> ansB; its potentials are dropped).  Both start from the potentials `v0`.
//
// Round 4 ran them one after the other as nested loops -- over the rows left without a column, over the search iterations of
// a row, over the columns of its augmenting path -- and a wave's envs walked those in lockstep: every loop ran as long as the
// env of the wave that needed it longest (tools/matching_sim.py: 42.5 search trips + 23.5 path trips per wave and step at ten
// agents, where one env needs 17.3 + 9.9).  Here every env advances its own state machine: a trip of the one loop is one
// search iteration of whatever row / matching the env is at, so the wave runs as long as its busiest env's TOTAL (28.3 trips),
// and the path is flipped in one parallel step instead of a walk: each column carries the bit mask of the columns on the path
// from the root to itself (`pmask`: the mask of the column it was reached from, plus itself), so when the search ends at a
// free column j1 the columns of pmask(j1) each take the row of their predecessor at once (one ds_bpermute).
// The arithmetic of one env is the same sequence as before: results and potentials are bit-identical.
__device__ __forceinline__ double hungarian_pair(const SegLanes &sl, const bool act, const double2 *x, const double2 PA, const double uA,
                                                 const int mineA, const bool needB, const double2 PB, const double uB, const int mineB,
                                                 const double v0, int8_t *ansA, int8_t *ansB, int *claim) {
    const int N = sl.N, li = sl.li;
    const double INF = 1e300;
    // start: the lowest row that claims a column gets it (those edges are tight); claim[0..N) for A, [N..2N) for B
    if (act) { claim[li] = N; claim[N + li] = N; }
    wave_sync();
    if (act) { atomicMin(&claim[mineA], li); if (needB) atomicMin(&claim[N + mineB], li); }
    wave_sync();
    int prow = -1, prowB = -1;   // row matched to column li
    bool unA = false, unB = false;
    if (act) {
        const int c = claim[li], cB = claim[N + li];
        prow = c < N ? c : -1;
        prowB = cB < N ? cB : -1;
        unA = claim[mineA] != li;
        unB = needB && claim[N + mineB] != li;
    }
    uint32_t um = sl.bits(unA), umB = sl.bits(unB);
#ifdef FMARL_HSTAT
    if (act && li == 0) {
        atomicAdd(&g_fmarl_hstat[0][0], 1ull); atomicAdd(&g_fmarl_hstat[0][2], (unsigned long long)__builtin_popcount(um));
        if (needB) { atomicAdd(&g_fmarl_hstat[1][0], 1ull); atomicAdd(&g_fmarl_hstat[1][2], (unsigned long long)__builtin_popcount(umB)); }
    }
#endif
    double2 Pc = PA;
    double u = uA, v = v0, vfin = v0, minv = INF;
    int pfin = prow, i = 0, i0 = 0, j0 = -1, way = -1, second = 0;
    uint32_t pmask = 0, pm0 = 0;
    bool usedc = false, in_tree = false, working = act;
    // the next row without a column; when the matching has none left: on to matching B (its potentials start from v0 again), then out
#define FMARL_NEXT_ROOT()                                                                                                    \
    {                                                                                                                        \
        if (um == 0 && second == 0) { vfin = v; pfin = prow; second = 1; Pc = PB; u = uB; v = v0; prow = prowB; um = umB; }  \
        if (um == 0) working = false;                                                                                        \
        else {                                                                                                               \
            i = __builtin_ctz(um); um &= um - 1;                                                                             \
            minv = INF; way = -1; usedc = false; in_tree = li == i; i0 = i; j0 = -1; pm0 = 0;                                \
        }                                                                                                                    \
    }
    FMARL_NEXT_ROOT()
    for (int trip = 0; working && trip < 2 * N * (N + 1); ++trip) {   // (a row's search adds at most N columns to its tree)
#ifdef FMARL_HSTAT
        if (li == 0) atomicAdd(&g_fmarl_hstat[second][3], 1ull);
#endif
        const double ui0 = bperm_f64(sl.at(i0), u);
        const bool open = !usedc;
        if (open) {
            const double cur = dist2(x[i0], Pc) - ui0 - v;
            if (cur < minv) { minv = cur; way = j0; pmask = pm0 | (1u << li); }
        }
        const double delta = sl.extreme<false>(open ? minv : INF);
        const uint32_t hit = sl.bits(open && minv == delta);   // ties to the lowest column
        const int j1 = __builtin_ctz(hit);
        if (in_tree) u += delta;
        if (usedc) v -= delta; else minv -= delta;
        if (li == j1) usedc = true;
        const int a1 = sl.at(j1);
        const int r1 = bperm_i32(a1, prow);
        const uint32_t pm1 = (uint32_t)bperm_i32(a1, (int)pmask);
        if (r1 < 0) {   // a free column: flip the augmenting path back to the root (row i), all its columns at once
            const int pw = bperm_i32(sl.at(way < 0 ? 0 : way), prow);
            if ((pm1 >> li) & 1u) prow = way < 0 ? i : pw;
            FMARL_NEXT_ROOT()
        } else {
            if (li == r1) in_tree = true;
            i0 = r1; j0 = j1; pm0 = pm1;
        }
    }
#undef FMARL_NEXT_ROOT
    double vout = 0.0;
    if (act) {
        ansA[pfin] = (int8_t)li;
        if (needB) ansB[prow] = (int8_t)li;
        vout = vfin - sl.extreme<true>(vfin);
    }
    return vout;
}

// Per-env LDS tables of the formation kernels, after the common ones (byte offsets in Params.f_*).
struct FormLds {
    char *base, *dead;   // the env's block / its part of the second region (tables nobody reads once the emission starts)
    const Params &p;
    __device__ FormLds(const Params &p_, char *lds, uint32_t el)
        : base(lds + (size_t)el * p_.lds_env_bytes), dead(lds + p_.lds_stage + (size_t)el * p_.lds2_bytes), p(p_) {}
    __device__ double2 *pos() const { return (double2 *)(base + p.lds_pos); }
    __device__ float2 *velf() const { return (float2 *)(base + p.lds_agentf); }     // (vx, vy) in f32
    __device__ float2 *posf() const { return (float2 *)(base + p.lds_posf); }
    __device__ double *wall() const { return (double *)(base + p.lds_wall); }
    __device__ int *flag() const { return (int *)(base + p.lds_flag); }
    __device__ double2 *slot_new() const { return (double2 *)(base + p.f_slot_new); }
    __device__ float2 *slotf() const { return (float2 *)(base + p.f_slotf); }       // (float)slot_new: what the node rows read
    __device__ const float4 *wallf() const { return (const float4 *)(base + p.lds_wallf); }   // (e0, axis + w/2, e1, axis - w/2) in f32
    __device__ double2 *slot_old() const { return (double2 *)(dead + p.f_slot_old); }
    // small per-agent indices as bytes (N <= 32): slot of the matching on the current / previous slots, nearest slot
    // within thr (-1: none), and at [3 N] the nearest previous slot of agent 0
    __device__ int8_t *g_new() const { return (int8_t *)(base + p.f_g); }
    __device__ int8_t *g_old() const { return (int8_t *)(base + p.f_g) + p.N; }
    __device__ int8_t *near_new() const { return (int8_t *)(base + p.f_g) + 2 * p.N; }
    __device__ int8_t *near_old0() const { return (int8_t *)(base + p.f_g) + 3 * p.N; }
    __device__ uint32_t *masks() const { return (uint32_t *)(base + p.f_masks); }   // [N][3]: b, flag, obs code
    __device__ double *theta() const { return (double *)(dead + p.f_theta); }
    __device__ double *vdual() const { return (double *)(dead + p.f_words); }   // column potentials the last matching left (state)
    __device__ uint32_t *words() const { return (uint32_t *)(base + p.lds_flag) + 1; }   // occ_old, occ_new, occ_final (behind the flag)
    __device__ double *stat() const { return (double *)(dead + p.lds_stat); }   // [pd_new | Dg_old] x N
    __device__ uint32_t *openmask() const { return (uint32_t *)(dead + p.lds_stat + 2 * p.N * 8); }   // bit j: agent j has not arrived yet
    __device__ bool skip() const { return *flag() != 0; }

    // goal of agent entity e as seen in the graph row of ego i (ff:916-943), as the float32 rounding the row starts from
    __device__ float2 graph_goal(uint32_t i, uint32_t e) const {
        const int nr = near_new()[e];
        if (nr >= 0) return slotf()[nr];
        if ((masks()[3 * i] >> e) & 1) return slotf()[g_new()[i]];
        return posf()[e];   // (its own position)
    }
};

// ff:518-530: wall box WITHOUT the 1.05 factors of navigation_graph
__device__ __forceinline__ bool wall_box_hit_plain(double2 x, double axis, double e0, double e1, int orient) {
    const double s = kEntitySize;
    const double pperp = orient == 0 ? x.y : x.x, ppar = orient == 0 ? x.x : x.y;
    return (axis - s / 2 <= pperp) && (pperp <= axis + s / 2) && (e0 - s / 2 <= ppar) && (ppar <= e1 + s / 2);
}

// One branch event of ff:707-739 / ff:916-943 on the occupancy mask.  Returns type (0 = near slot,
// 1 = Hungarian slot of `ego`, 2 = own position) in bits 1..2 and the observed flag in bit 0.
__device__ __forceinline__ uint32_t branch_event(int near_e, int g_ego, uint32_t full, uint32_t &occ) {
    if (near_e >= 0) { occ |= 1u << near_e; return 1u; }
    if ((~occ) & full) return 2u | ((occ >> g_ego) & 1u);
    occ = 0;
    return 4u;
}

constexpr int kFormationRecordWords = 9;   // include/fmarl.h fmarl_step_record_words

// One node_obs row (ff:896-971) of the wave's envs: row q = (env, ego a, entity e) -> F = 12 floats as three 16-byte chunks
// [dv dx] [goal flag dx.x] [dx.y dx type]; the chunks share the position loads and the index math.
__device__ __forceinline__ void formation_row(const Params &p, char *lds, int el0w, uint32_t q, float4 &c0, float4 &c1, float4 &c2) {
    const uint32_t N = p.N, NE = N * p.E, first_wall = N + p.L + p.O;
    const uint32_t e_l = p.dNE.div(q), r = q - e_l * NE;
    const FormLds te(p, lds, el0w + e_l);
    const uint32_t a = p.dE.div(r), e = r - a * p.E;
    // differences of the f32 roundings (as navigation_graph's rows): within 1.2e-7 of the rounded f64 difference
    const float2 vi = te.velf()[a], pi = te.posf()[a], pe = te.posf()[e];
    const float dx = pe.x - pi.x, dy = pe.y - pi.y;
    float vx = 0.f, vy = 0.f, gx = dx, gy = dy, fl = 1.f, t7 = dx, t8 = dy, t9 = dx, t10 = dy;
    if (e < N) {
        const float2 ve = te.velf()[e];
        vx = ve.x; vy = ve.y;
        const float2 gl = te.graph_goal(a, e);
        gx = gl.x - pi.x; gy = gl.y - pi.y;
        fl = (float)((te.masks()[3 * a + 1] >> e) & 1);
    } else if (e >= first_wall) {
        const float4 wc = te.wallf()[e - first_wall];   // corners (e0, axis + w/2), (e1, axis - w/2): ff:963-964
        t7 = wc.x - pi.x; t8 = wc.y - pi.y; t9 = wc.z - pi.x; t10 = wc.w - pi.y;
    }
    const float type = e < N ? 0.f : (e < N + p.L ? 1.f : (e < first_wall ? 2.f : 3.f));
    c0 = make_float4(vx - vi.x, vy - vi.y, dx, dy);
    c1 = make_float4(gx, gy, fl, t7);
    c2 = make_float4(t8, t9, t10, type);
}

// `nrows` consecutive 48-byte rows (lane l holds row l as three chunks) to gdst through the wave's LDS window, so that global
// memory sees contiguous 16-byte chunks -- 1 KiB per store instruction -- instead of 16 bytes per lane at a 48-byte stride
// (three times the write requests for the same bytes).  The window is the wave's part of the second LDS region: its envs'
// tables there are dead by now, and it is private to the wave (wave-local ordering).  The frame starts at the previous
// 64-byte boundary of gdst (rows are 16-byte aligned, so the offset is whole chunks): lane quads then write whole blocks.
__device__ __forceinline__ void formation_flush_rows(const Params &p, char *lds, const float4 &c0, const float4 &c1, const float4 &c2, uint32_t nrows,
                                                     float4 *gdst) {
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 *buf = (float4 *)(lds + p.lds_stage + wave * p.stage_wave_bytes);
    const uint32_t shift = (uint32_t)(((uintptr_t)gdst >> 4) & 3), end = shift + 3 * nrows;
    if (lane < nrows) { buf[shift + 3 * lane] = c0; buf[shift + 3 * lane + 1] = c1; buf[shift + 3 * lane + 2] = c2; }
    wave_sync();
    float4 *gal = gdst - shift;
    // all LDS reads first, then the stores (at most 3 chunks per lane: 3 * 63 + 3 <= 192)
    const bool in0 = lane >= shift && lane < end, in1 = lane + 64 < end, in2 = lane + 128 < end;
    float4 v0, v1, v2;   // (each only read under the predicate it was loaded under: no initialisers -- they were twelve moves per window)
    if (in0) v0 = buf[lane];
    if (in1) v1 = buf[lane + 64];
    if (in2) v2 = buf[lane + 128];
    if (in0) gal[lane] = v0;
    if (in1) gal[lane + 64] = v1;
    if (in2) gal[lane + 128] = v2;
    wave_sync();   // the next window's writes stay behind these reads
}

// node_obs rows of the envs [el0w, el0w + nenv_w) of the workgroup by one wave (ff:896-971): shared by the step / reset
// kernels and the learner-side rebuild (formation_rebuild_kernel), which fills the same LDS tables from the records.
__device__ __forceinline__ void formation_emit_rows(const Params &p, const FmarlOutputs &o, char *lds, int env0, int el0w,
                                                    int nenv_w, int lane) {
    const int N = p.N;
    if (!o.node_obs) return;
    const uint32_t NE = N * p.E, total = nenv_w * NE;
    float4 *dst = (float4 *)(o.node_obs + ((size_t)env0 + el0w) * NE * 12);
    const bool some_skip = __ballot(lane < nenv_w && FormLds(p, lds, el0w + lane).skip()) != 0;
    wave_sync();   // every lane is done with the tables of the second region: it becomes the wave's window
    if (!some_skip && p.f_rows >= 8) {
        const uint32_t R = p.f_rows;
        for (uint32_t w0 = 0; w0 < total; w0 += R) {
            const uint32_t nrows = min(R, total - w0);
            float4 c0, c1, c2;   // (lanes beyond nrows never write theirs to the window)
            if ((uint32_t)lane < nrows) formation_row(p, lds, el0w, w0 + lane, c0, c1, c2);
            formation_flush_rows(p, lds, c0, c1, c2, nrows, dst + (size_t)w0 * 3);
        }
    } else {   // some envs keep their previous rows (reset in flight): per-lane stores of the others
        for (uint32_t q = lane; q < total; q += 64) {
            if (FormLds(p, lds, el0w + p.dNE.div(q)).skip()) continue;
            float4 c0, c1, c2;
            formation_row(p, lds, el0w, q, c0, c1, c2);
            float4 *d = dst + (size_t)q * 3;
            d[0] = c0; d[1] = c1; d[2] = c2;
        }
    }
}

// mean and population std (two-pass, like np.mean / np.std) of the dists_to_goal vector seen by the agent loop: entry j
// is this step's value for j < split -- the path length pd[j] while agent j is still under way (bit j of open), else
// the value frozen at its arrival -- and the previous step's value (stale[j]) for j >= split.
__device__ __forceinline__ void travelled_stats(const double *pd, const double *stale, uint32_t open, int n, int split,
                                                double &mean, double &sd, double &m2) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += (j < split && ((open >> j) & 1u)) ? pd[j] : stale[j];
    mean = s / n;
    double q = 0.0;
    for (int j = 0; j < n; ++j) {
        const double d = ((j < split && ((open >> j) & 1u)) ? pd[j] : stale[j]) - mean;
        q += d * d;
    }
    m2 = q;
    sd = sqrt_pos(q / n);
}
// What one agent's thread carries from one step of a span to the next (formation_span_kernel): everything a step reads at its
// head.  `carry` bit 0 = this step's state arrives here (left by the previous step of the span; the static entities are still in
// the envs' LDS tables) -- nothing is loaded but the action; bit 1 = the new state stays here instead of going to global memory.
struct FormCarry { double2 x, v, so; double pd, vd, Dg, Tr; int noc, nac, step; bool socc, fdone; };   // (constants -- rot, min_time -- are re-read: cached loads)

// STEP = true : one env step (MultiAgentGraphEnv.step, environment.py:816-877)
// STEP = false: observation of freshly reset envs (MultiAgentGraphEnv.reset, environment.py:892-897)
template <bool STEP>
__device__ __forceinline__ void formation_body(const Params &p, const FmarlOutputs &o, const int32_t *action_idx,
                                               const float *action_vec, int auto_reset, FormCarry &c, const int carry) {
    const bool tables_loaded = (carry & 5) != 0;   // (bit 2: the tables only -- the measure builds' span without the register carry)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    FMARL_TICKS_BEGIN
    const int tid = threadIdx.x, N = p.N;
    const int env0 = env_block(p) * p.epb;
    const int nenv = min(p.epb, p.n_envs - env0);
    // Every env lives inside ONE wave (p.epw = 64 / N envs per wave, p.epb = 4 p.epw): after the shared entity tables are
    // loaded no wave ever waits for another one -- all the ordering between the phases below is wave-local
    // (wave_sync: LDS executes a wave's instructions in order, a fence keeps the compiler from reordering them).
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int elw = lane / N, i = lane - elw * N;
    const int el0w = wave * p.epw;                            // first env of this wave
    const int nenv_w = max(0, min(p.epw, nenv - el0w));      // envs of this wave
    const int el = el0w + elw;
    const bool active = elw < nenv_w;
    const int env = env0 + el;
    const size_t g = (size_t)env * N + i;
    const FormLds t(p, lds, active ? el : 0);
    double *s_stat = t.stat();   // [pd_new | Dg_old] x N, then the mask of agents still under way
    const uint32_t full = N >= 32 ? ~0u : ((1u << N) - 1);

    double2 x = make_double2(0, 0), v = make_double2(0, 0);
    double pd = 0;
    int step = 0;
    bool emit = false;   // this env's obs / node_obs / adj are written by this launch
    // Every global load of the launch is issued here, in one batch: loads placed where their value is first used cost a
    // memory round trip each (nine in a row at the head of every wave), and a load behind stores waits for those stores
    // (one vmcnt counter orders both on gfx9).
    double Dg_old = 0, Tr_old = 0, fdone = 0, socc = 0, mtime = 0, vd = 0;
    double2 so = make_double2(0, 0), rot = make_double2(1, 0);
    int noc_old = 0, nac_old = 0, a_pre = -1, cs = 0, rf = 0;
    if (STEP && (carry & 1)) {
        x = c.x; v = c.v; pd = c.pd; so = c.so; vd = c.vd; socc = c.socc ? 1.0 : 0.0; cs = c.step;
        Dg_old = c.Dg; Tr_old = c.Tr; fdone = c.fdone ? 1.0 : 0.0; noc_old = c.noc; nac_old = c.nac;
        if (active) {
            rot = p.rot_table[i];
            if (o.info) mtime = p.min_time[g];
            if (action_idx) a_pre = action_idx[g];
        }
    } else if (active) {
        x = p.agent_pos[g]; v = p.agent_vel[g]; pd = p.p_dist[g];
        so = p.slot_pos[g];
        if (STEP) rot = p.rot_table[i];
        vd = p.match_dual[g];
        socc = p.slot_occ[g];
        cs = p.cur_step[env];
        if (!STEP) rf = p.reset_flag[env];
        if (STEP) {
            Dg_old = p.dists_to_goal[g]; Tr_old = p.times_required[g]; fdone = p.formation_done[g];
            noc_old = p.num_obst_coll[g]; nac_old = p.num_agent_coll[g];
            if (o.info) mtime = p.min_time[g];
            if (action_idx) a_pre = action_idx[g];
        }
    }
    // reset observation: workgroups without a freshly reset env have nothing to do (block-uniform exit)
    if (!STEP && !__syncthreads_or(active && rf != 0)) return;
    // (a later step of a span: the static entities are still in the envs' tables -- no episode ends inside a span)
    if (!tables_loaded) load_statics_range(p, lds, env0, 0, nenv, tid, blockDim.x);   // (its loads join the batch: nothing above has waited for a value yet)
    if (active) {
        t.pos()[i] = x;
        t.slot_old()[i] = so;
        t.vdual()[i] = vd;
        if (i == 0) { t.words()[0] = 0; t.words()[1] = 0; *t.openmask() = 0; }
        step = cs + (STEP ? 1 : 0);
        emit = STEP ? !(auto_reset && step >= p.episode_length) : rf != 0;
        if (i == 0) *t.flag() = emit ? 0 : 1;
    }
    // the only workgroup barrier: the entity tables are loaded by all four waves together (a later step of a span has none:
    // every env lives in one wave, and nothing a wave reads was written by another one)
    if (!tables_loaded) __syncthreads();
    else wave_sync();
    if (active && socc != 0.0) atomicOr(&t.words()[0], 1u << i);   // occupancy left by the previous pass
    FMARL_TICK(0);   // loads issued, entity tables, the barrier

    if (STEP && active) world_step_agent(p, t.base, i, g, action_idx, action_vec, x, v, pd, true, a_pre);
    wave_sync();   // every lane has finished reading the old positions
    FMARL_TICK(1);   // physics

    double Tr_new = 0;
    const double2 L0 = active ? t.pos()[N] : make_double2(0, 0);   // landmark 0
    if (active) {
        t.pos()[i] = x;
        t.velf()[i] = make_float2((float)v.x, (float)v.y);
        t.posf()[i] = make_float2((float)x.x, (float)x.y);
        if (STEP) {
            // ff:35-40 find_angle = arctan2 wrapped to [0, 2 pi); only its ORDER over the agents is needed (the anchor of the ring
            // is the agent with the smallest angle, ff:633-636): a monotone key without the arctangent -- quadrant + |dy| /
            // (|dx| + |dy|).  Two agents whose angles agree to the last bits may swap: the ring then turns by that much.
            const double dx = x.x - L0.x, dy = x.y - L0.y, ax = fabs(dx), ay = fabs(dy), sum = ax + ay;
            const double frac = sum > 0.0 ? ay * rcp_nr(sum) : 0.0;
            t.theta()[i] = dy >= 0.0 ? (dx > 0.0 || sum == 0.0 ? frac : 2.0 - frac) : (dx < 0.0 ? 2.0 + frac : 4.0 - frac);
            const bool open = Tr_old == -1.0;
            const double fd = sqrt_pos(dx * dx + dy * dy);   // ff:445-451 ring test
            const bool ring = fd < 1.05 * kTargetRadius && fd > 0.95 * kTargetRadius;
            if (ring) fdone = 1.0;
            Tr_new = (ring && open) ? step * kDt : Tr_old;
            s_stat[i] = pd; s_stat[N + i] = Dg_old;
            if (open) atomicOr(t.openmask(), 1u << i);
        }
    }
    wave_sync();

    double2 P = make_double2(0, 0);   // this agent's own slot index i of the current ring (column i of the matching)
    if (active) {   // slots: ff:630-648 (step: inside reward(agent 0)); on reset they come from the state
        if (STEP) {
            // slot i = landmark 0 + radius (cos, sin)(theta_min + i 2 pi / N): the anchor agent's own direction (dx, dy) / r
            // is (cos, sin)(theta_min); turning it by the tabulated (cos, sin)(i 2 pi / N) needs no trigonometric call
            double kmin = t.theta()[0];
            int jmin = 0;
            for (int j = 1; j < N; ++j) { const double kj = t.theta()[j]; if (kj < kmin) { kmin = kj; jmin = j; } }
            const double2 xa = t.pos()[jmin];
            const double dx = xa.x - L0.x, dy = xa.y - L0.y, r2 = dx * dx + dy * dy;
            double inv_r = 0.0;
            if (r2 > 0.0) (void)sqrt_inv_pos(r2, inv_r);
            const double c0 = r2 > 0.0 ? dx * inv_r : 1.0, s0 = r2 > 0.0 ? dy * inv_r : 0.0;   // arctan2(0, 0) = 0
            P = make_double2(L0.x + kTargetRadius * (c0 * rot.x - s0 * rot.y), L0.y + kTargetRadius * (s0 * rot.x + c0 * rot.y));
        } else {
            P = t.slot_old()[i];
        }
        t.slot_new()[i] = P;
        t.slotf()[i] = make_float2((float)P.x, (float)P.y);
    }
    wave_sync();
    FMARL_TICK(2);   // angle keys, ring test, slots

    double left = 0;
    // start of the matchings (hungarian_seg): this row's reduced minimum and its column, against the current / the previous slots
    double rbest = 1e300, rbest_old = 1e300;
    int rkb = 0, rkb_old = 0;
    if (active) {   // agent x slot distances (ff:650-655), nearest slot within thr, dist_left (ff:453)
        // row minima of c - v for the warm-started matchings (v: the potentials the previous matching of this env left;
        // for the previous slots they are the potentials of exactly those slots, one step of motion ago)
        const double *vg = t.vdual();   // (staged in LDS at kernel start: a global load per loop trip would stall every trip)
        double best = 1e300, best_old = 1e300;
        int kb = 0, kb_old = 0;
        for (int k = 0; k < N; ++k) {
            const double vk = vg[k];
            const double d = dist2(x, t.slot_new()[k]);
            if (d < best) { best = d; kb = k; }
            if (d - vk < rbest) { rbest = d - vk; rkb = k; }
            if (STEP) {
                const double d0 = dist2(x, t.slot_old()[k]);
                if (d0 < best_old) { best_old = d0; kb_old = k; }
                if (d0 - vk < rbest_old) { rbest_old = d0 - vk; rkb_old = k; }
            }
        }
        left = best;
        t.near_new()[i] = (int8_t)(best < p.thr ? kb : -1);
        if (i == 0) *t.near_old0() = (int8_t)(STEP ? (best_old < p.thr ? kb_old : -1) : (best < p.thr ? kb : -1));
    }
    wave_sync();
    FMARL_TICK(3);   // agent x slot distances

    if (active) {
        // occupancy recomputed in reward(agent 0): any agent within thr of slot i (ff:660-661)
        bool occ = false;
        const double2 Pi = t.slot_new()[i];
        for (int a = 0; a < N; ++a) occ |= closer_than(t.pos()[a], Pi, p.thr);   // decided on the squares unless within an ulp of thr
        if (occ) atomicOr(&t.words()[1], 1u << i);
    }
    wave_sync();
    FMARL_TICK(4);   // occupancy
    double vd_new = 0.0;   // column potential the matching on the current slots leaves: the next step's warm start
    if (!FMARL_SKIP(p, 64)) {
        // the matchings, every env of the wave at once on its own agent lanes: against the current slots, then (a step) against
        // the previous ones -- that one only serves observation(agent 0) in its "free slot left" branch (ff:707-739): not when
        // agent 0 sits on a previous slot or every slot is taken (common once agents hold the ring)
        const SegLanes sl(N, i, elw * N);
        int *claim = (int *)t.vdual();   // (the potentials' table, N doubles = 2 N ints: every lane has its own entry in `vd`, the distance pass is through)
        const bool need = STEP && active && !(*t.near_old0() >= 0 || ((~t.words()[0]) & full) == 0);
#ifdef FMARL_HSTAT
        if (STEP && active && !need && i == 0) atomicAdd(&g_fmarl_hstat[1][1], 1ull);
#endif
        vd_new = hungarian_pair(sl, active, t.pos(), P, rbest, rkb, need, so, rbest_old, rkb_old, vd, t.g_new(), t.g_old(), claim);
    }
    wave_sync();
    FMARL_TICK(5);   // matchings

    // Entity sets of the walk below, built by the agent lanes in parallel (the potentials' LDS table is free again):
    // sm[k] = agent entities sitting near slot k, sm[N] = all entities near some slot, sm[N + 1] = the slots they cover.
    if (active) {
        uint32_t *sm = (uint32_t *)t.vdual();
        sm[i] = 0;
        if (i == 0) { sm[N] = 0; sm[N + 1] = 0; }   // (N = 1 has no second lane)
    }
    wave_sync();
    if (active) {
        const int ne = t.near_new()[i];
        if (ne >= 0) {
            uint32_t *sm = (uint32_t *)t.vdual();
            atomicOr(&sm[ne], 1u << i); atomicOr(&sm[N], 1u << i); atomicOr(&sm[N + 1], 1u << ne);
        }
    }
    wave_sync();

    if (lane < nenv_w && !FMARL_SKIP(p, 128)) {
        // Walk of the occupancy mask in the reference's call order, one lane per env (the wave's envs side by side in its
        // first lanes): for every ego a, observation(a)'s event, then the N row events of graph_observation(a).  The row
        // events of one ego have a closed form unless one of them can find every slot taken: near entities only add
        // their slot, and a non-near entity e reads bit g_a of (S | slots of the near entities before e).  Only when
        // S | (all near slots) is full -- agents holding the whole ring -- the events are walked one by one.
        const FormLds tw(p, lds, el0w + lane);
        uint32_t occ = tw.words()[0];
        uint32_t *m = tw.masks();
        const int8_t *gn = tw.g_new(), *nr = tw.near_new();
        const uint32_t *sm = (const uint32_t *)tw.vdual();
        const uint32_t NEAR = sm[N], PN = sm[N + 1], NN = full & ~NEAR;
        const int near0 = *tw.near_old0();
        // Ego 0 (whose observation reads the PREVIOUS slots), then the question whether the other egos depend on each other at all:
        // once every near slot is in the mask (ego 0's row events put them there) and the mask is not full, nothing an ego a >= 1
        // does changes it -- its near slot is in it already, and nobody can find every slot taken -- so every ego reads the same
        // mask and the agent lanes evaluate theirs side by side below (`fast`: the common case; this serial walk was 5 % of a
        // step on six lanes of a wave).  Otherwise the walk goes on one ego after the other, as before.
        bool fast = false;
        for (int a = 0; a < N; ++a) {
            uint32_t code;
            if (STEP && a == 0) {
                code = branch_event(near0, tw.g_old()[0], full, occ);   // observation(0): previous slots
                occ = tw.words()[1];                                     // reward(0)
            } else {
                code = branch_event(a == 0 ? near0 : nr[a], gn[a], full, occ);
            }
            uint32_t mb, mf;
            const int ga = gn[a];
            if ((occ | PN) != full || NN == 0) {   // nobody can find every slot taken in this pass
                const uint32_t smg = sm[ga];       // entities near the ego's matched slot: after the first one, bit g_a is set
                const uint32_t later = smg ? ~((2u << __builtin_ctz(smg)) - 1u) : 0u;
                mb = NN;
                mf = NEAR | (NN & (((occ >> ga) & 1u) ? full : later));
                occ |= PN;
            } else {
                mb = 0; mf = 0;
                for (int e = 0; e < N; ++e) {
                    const uint32_t c = branch_event(nr[e], ga, full, occ);
                    mb |= ((c >> 1) & 1u) << e;
                    mf |= (c & 1u) << e;
                }
            }
            m[3 * a] = mb; m[3 * a + 1] = mf; m[3 * a + 2] = code;
            if (a == 0 && N >= 3 && (occ & PN) == PN && occ != full) { fast = true; break; }
        }
        tw.words()[2] = occ;
        if (N >= 3) ((uint32_t *)tw.vdual())[N + 2] = fast ? 1u : 0u;   // (the table has N doubles = 2 N words: room for word N + 2 from three agents on; `fast` is never set below)
    }
    wave_sync();
    if (active && i >= 1 && N >= 3 && ((const uint32_t *)t.vdual())[N + 2] != 0 && !FMARL_SKIP(p, 128)) {
        // ego i on the mask ego 0 left (branch_event's near / free-slot cases, then the closed form of its N row events)
        const uint32_t *sm = (const uint32_t *)t.vdual();
        const uint32_t occ1 = t.words()[2], NEAR = sm[N], NN = full & ~NEAR;
        const int nri = t.near_new()[i], ga = t.g_new()[i];
        const uint32_t smg = sm[ga];
        const uint32_t later = smg ? ~((2u << __builtin_ctz(smg)) - 1u) : 0u;
        uint32_t *m = t.masks();
        m[3 * i] = NN;
        m[3 * i + 1] = NEAR | (NN & (((occ1 >> ga) & 1u) ? full : later));
        m[3 * i + 2] = nri >= 0 ? 1u : (2u | ((occ1 >> ga) & 1u));
    }
    wave_sync();
    FMARL_TICK(6);   // entity sets + walk

    if (active) {
        const uint32_t code = t.masks()[3 * i + 2];
        const bool old_slots = STEP && i == 0;
        const int nr = i == 0 ? (int)*t.near_old0() : (int)t.near_new()[i];
        double2 goal = x;   // type 2
        if (code & 1u && !(code & 6u)) goal = (old_slots ? t.slot_old() : t.slot_new())[nr];                 // type 0
        else if (code & 2u) goal = old_slots ? t.slot_old()[t.g_old()[0]] : t.slot_new()[t.g_new()[i]];     // type 1
        const double flag = (double)(code & 1u);
        if (o.obs && emit) {   // ff:740-741: concat(v, x, goal - x) + flag
            float *ob = o.obs + g * p.D;
            ob[0] = (float)(v.x + flag); ob[1] = (float)(v.y + flag); ob[2] = (float)(x.x + flag); ob[3] = (float)(x.y + flag);
            ob[4] = (float)(goal.x - x.x + flag); ob[5] = (float)(goal.y - x.y + flag);
        }
        const bool keep = STEP && (carry & 2) != 0;   // a span's inner step: the new state stays in registers
        if (keep) {
            c.socc = ((t.words()[2] >> i) & 1u) != 0;
            c.vd = vd_new;
        } else if (STEP || emit) {
            p.slot_occ[g] = (double)((t.words()[2] >> i) & 1u);
            p.match_dual[g] = vd_new;   // column potentials of the matching on the current slots: next step's warm start
        }
        if (o.graph_record && emit) {   // what a learner on another GPU needs to rebuild this env's node_obs (fmarl.h)
            uint32_t *r = o.graph_record + g * kFormationRecordWords;
            const float2 pf = t.posf()[i], vf = t.velf()[i];
            const double2 sl = t.slot_new()[i];
            r[0] = __float_as_uint(pf.x); r[1] = __float_as_uint(pf.y); r[2] = __float_as_uint(vf.x); r[3] = __float_as_uint(vf.y);
            r[4] = __float_as_uint((float)sl.x); r[5] = __float_as_uint((float)sl.y);
            r[6] = t.masks()[3 * i]; r[7] = t.masks()[3 * i + 1];
            r[8] = (uint32_t)(uint8_t)t.near_new()[i] | ((uint32_t)(uint8_t)t.g_new()[i] << 8);
        }
        FMARL_TICK(7);   // obs, record, potentials
        if (STEP) {
            const bool open = Tr_old == -1.0;
            const double Dg_new = open ? pd : Dg_old;
            const double delta = dist2(x, t.slot_new()[t.g_new()[i]]);   // ff:665
            double fairness, m, sd, m2 = 0.0;   // ff:623-628, same stale/fresh rule as navigation_graph
            const bool base = Dg_old != -1.0;   // (m, m2) describe the vector the info statistics differ from in this agent's entry
            if (!base) mixed_stats(s_stat, s_stat, N, N, m, sd);
            else travelled_stats(s_stat, s_stat + N, *t.openmask(), N, i, m, sd, m2);
            fairness = ratio_out(m, sd + 0.0001);
            int ag_hits = 0;
            for (int j = 0; j < N; ++j)
                if (j != i && closer_than(x, t.pos()[j], 1.05 * (kEntitySize + kEntitySize))) ++ag_hits;
            bool ob_hit = false;
            for (int k = 0; k < p.O; ++k)
                ob_hit |= closer_than(t.pos()[N + p.L + k], x, 1.05 * (kEntitySize + kEntitySize));
            const double *wl = t.wall();
            for (int w = 0; w < p.W; ++w)
                ob_hit |= wall_box_hit_plain(x, wl[w * 4], wl[w * 4 + 1], wl[w * 4 + 2], (int)wl[w * 4 + 3]);
            double rew = delta < p.thr ? p.goal_rew : -delta;   // ff:668-699
            rew -= p.collision_rew * ag_hits;
            if (ob_hit) rew -= p.collision_rew;
            rew += p.fair_rew * tanh_out(fairness - 5.0);
            rew = fmin(fmax(rew, -2 * p.collision_rew), p.goal_rew + p.fair_rew);

            FMARL_TICK(8);   // statistics, hits, reward
            const int noc = noc_old + (ob_hit ? 1 : 0), nac = nac_old + ag_hits;
            if (keep) {
                c.x = x; c.v = v; c.pd = pd; c.Dg = Dg_new; c.Tr = Tr_new; c.noc = noc; c.nac = nac;
                c.so = t.slot_new()[i]; c.fdone = fdone != 0.0; c.step = step;
            } else {
                p.agent_pos[g] = x; p.agent_vel[g] = v; p.p_dist[g] = pd;
                p.dists_to_goal[g] = Dg_new; p.times_required[g] = Tr_new; p.dist_left[g] = left;
                p.num_obst_coll[g] = noc; p.num_agent_coll[g] = nac;
                p.slot_pos[g] = t.slot_new()[i]; p.slot_delta[g] = delta; p.formation_done[g] = fdone;
                if (i == 0) p.cur_step[env] = step;
            }
            if (o.reward) o.reward[g] = (float)rew;
            if (o.done) o.done[g] = step >= p.episode_length;
            FMARL_TICK(9);   // state stores
            if (o.info) {   // ff:477-499
                // ff:477-499: the same vector with this agent's own entry fresh -- one entry replaced (its path length for the frozen
                // value, if it is still under way), else unchanged
                double dm = m, ds = sd, unused;
                if (!base || (open && !replaced_entry_stats(m, m2, N, Dg_old, pd, dm, ds)))
                    travelled_stats(s_stat, s_stat + N, *t.openmask(), N, i + 1, dm, ds, unused);
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
                inf[FMARL_INFO_TIME_TAKEN * plane] = 0.f;                      // never updated by this scenario
                inf[FMARL_INFO_FORMATION_DIST * plane] = (float)fdone;          // 'Formation_dist' (ff:495)
                inf[FMARL_INFO_TIME_STDDEV * plane] = 0.f;
                inf[FMARL_INFO_TIME_MEAN_BY_STDDEV * plane] = 0.f;
                inf[FMARL_INFO_MIN_TIME_TO_GOAL * plane] = (float)mtime;
                inf[FMARL_INFO_INDIVIDUAL_REWARD * plane] = (float)rew;
            }
        }
    }
    FMARL_TICK(10);   // info planes
    // ---- emission: node_obs (16 bytes per lane) and adj
    if (FMARL_SKIP(p, 32)) { FMARL_TICKS_END; return; }
    formation_emit_rows(p, o, lds, env0, el0w, nenv_w, lane);
    FMARL_TICK(11);   // node rows
    emit_adj(p, o, lds, env0, el0w, el0w + nenv_w, lane, 64);
    FMARL_TICK(12);   // adj
    FMARL_TICKS_END;
}

#ifndef FMARL_FORM_MIN_BLOCKS
#define FMARL_FORM_MIN_BLOCKS 1
#endif
// SH = 1: BASELINE config 4's shape -- 10 agents, 1 landmark, 3 obstacles, the scenario's 2 walls -- as compile-time constants (fmarl_step.hip
// shape_const's reasoning); 0 = the run-time values.
template <int SH>
__device__ __forceinline__ void formation_shape_const(Params &q) {
    if constexpr (SH == 1) { q.N = 10; q.L = 1; q.O = 3; q.W = 2; q.E = 16; }
}
template <bool STEP, int SH>
__global__ __launch_bounds__(kThreads, FMARL_FORM_MIN_BLOCKS) void formation_kernel(Params p, FmarlOutputs o, const int32_t *action_idx,
                                                             const float *action_vec, int auto_reset) {
    FormCarry c;
    formation_shape_const<SH>(p);
    formation_body<STEP>(p, o, action_idx, action_vec, auto_reset, c, 0);
}

// fmarl_step_span for fair_graph_formation: T steps of the workgroup's own envs in one launch (no episode ends inside).
// Between the steps the agent's state stays in registers (FormCarry) and the static entities in the LDS tables: only the first
// step loads the state, only the last one stores it; every env lives inside one wave, so wave-local ordering is all the steps
// need (the next step's first LDS writes come behind this step's last reads of the emission window, in the wave's own order).
#ifndef FMARL_FORM_SPAN_BLOCKS
#define FMARL_FORM_SPAN_BLOCKS 4
#endif
template <int SH>
__global__ __launch_bounds__(kThreads, FMARL_FORM_SPAN_BLOCKS) void formation_span_kernel(Params p, FmarlOutputs o, SpanStrides s, const int32_t *action_idx,
                                                                     const float *action_vec, int T) {
    FormCarry c = {};
    for (int t = 0; t < T; ++t) {
        const Params &q = span_params_reloaded();   // == p
        // (the shape's counts: the argument block is re-read where it is used, so they are told to the compiler rather than written over a copy)
        if constexpr (SH == 1) __builtin_assume(q.N == 10 && q.L == 1 && q.O == 3 && q.W == 2 && q.E == 16);
        const FmarlOutputs ot = span_outputs(o, s, t);
#ifdef FMARL_FORM_NO_CARRY   // (A/B builds: the state through global memory every step, as before round 4)
        formation_body<true>(q, ot, action_idx ? action_idx + (size_t)t * s.actions : nullptr,
                             action_vec ? action_vec + (size_t)t * s.actions : nullptr, 0, c, t > 0 ? 4 : 0);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#else
        formation_body<true>(q, ot, action_idx ? action_idx + (size_t)t * s.actions : nullptr,
                             action_vec ? action_vec + (size_t)t * s.actions : nullptr, 0, c, (t > 0 ? 1 : 0) | (t < T - 1 ? 2 : 0));
        wave_sync();
#endif
    }
}

// Learner-side reconstruction of node_obs / adj of fair_graph_formation envs from the gathered records: the per-step
// record written by formation_kernel (FmarlOutputs.graph_record) and the once-per-episode record of the static entities
// (fmarl_rebuild.hip layout).  The LDS tables are filled with the float32 values the sender's emission read, then the
// same emission code runs: node_obs and adj equal the sender's bit for bit.  n_envs is the caller's.
__global__ __launch_bounds__(kThreads) void formation_rebuild_kernel(Params p, FmarlOutputs o, const uint32_t *ep_rec,
                                                                     const uint32_t *step_rec, int n_envs) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, N = p.N;
    const int env0 = env_block(p) * p.epb;
    const int nenv = min(p.epb, n_envs - env0);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int elw = lane / N, i = lane - elw * N;
    const int el0w = wave * p.epw;
    const int nenv_w = max(0, min(p.epw, nenv - el0w));
    const int el = el0w + elw;
    const int LO = p.L + p.O, words = 2 * N + 2 * LO + 6 * p.W;   // episode_record_words (fmarl_rebuild.hip)
    if (elw < nenv_w) {
        const FormLds t(p, lds, el);
        const uint32_t *r = step_rec + ((size_t)(env0 + el) * N + i) * kFormationRecordWords;
        const float2 pf = make_float2(__uint_as_float(r[0]), __uint_as_float(r[1]));
        t.posf()[i] = pf; t.pos()[i] = make_double2((double)pf.x, (double)pf.y);
        t.velf()[i] = make_float2(__uint_as_float(r[2]), __uint_as_float(r[3]));
        t.slot_new()[i] = make_double2((double)__uint_as_float(r[4]), (double)__uint_as_float(r[5]));
        t.slotf()[i] = make_float2(__uint_as_float(r[4]), __uint_as_float(r[5]));
        t.masks()[3 * i] = r[6]; t.masks()[3 * i + 1] = r[7]; t.masks()[3 * i + 2] = 0;
        t.near_new()[i] = (int8_t)(r[8] & 0xff); t.g_new()[i] = (int8_t)((r[8] >> 8) & 0xff);
        if (i == 0) *t.flag() = 0;
    }
    for (int k = tid; k < nenv * LO; k += blockDim.x) {
        const int e_l = k / LO, j = k - e_l * LO;
        const float *sp = (const float *)(ep_rec + (size_t)(env0 + e_l) * words) + 2 * (N + j);
        const FormLds t(p, lds, e_l);
        t.pos()[N + j] = make_double2((double)sp[0], (double)sp[1]);
        t.posf()[N + j] = make_float2(sp[0], sp[1]);
    }
    for (int k = tid; k < nenv * p.W; k += blockDim.x) {
        const int e_l = k / p.W, w = k - e_l * p.W;
        const FormLds t(p, lds, e_l);
        const uint32_t *q = ep_rec + (size_t)(env0 + e_l) * words + 2 * (N + LO) + 6 * w;
        const double axis = __longlong_as_double((long long)((unsigned long long)q[0] | ((unsigned long long)q[1] << 32)));
        const float *qf = (const float *)q;
        double *wl = t.wall() + w * 4;   // (float)e0 / (float)e1 travel: the sender's corner words are their float32 roundings too
        wl[0] = axis; wl[1] = (double)qf[2]; wl[2] = (double)qf[3]; wl[3] = (double)qf[4];
        ((float4 *)(t.base + p.lds_wallf))[w] = make_float4(qf[2], (float)(axis + kWallWidth / 2), qf[3], (float)(axis - kWallWidth / 2));
        t.pos()[N + LO + w] = qf[4] == 0.f ? make_double2(0.0, axis) : make_double2(axis, 0.0);
        t.posf()[N + LO + w] = qf[4] == 0.f ? make_float2(0.f, (float)axis) : make_float2((float)axis, 0.f);
    }
    __syncthreads();
    formation_emit_rows(p, o, lds, env0, el0w, nenv_w, lane);
    emit_adj(p, o, lds, env0, el0w, el0w + nenv_w, lane, 64);
}

}  // namespace fmarl
