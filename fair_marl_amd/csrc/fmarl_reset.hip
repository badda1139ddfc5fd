// Environment reset on the device (gfx950).
//
// Restates reference multiagent/custom_scenarios/navigation_graph.py:212-575 (reset_world +
// random_scenario) and multiagent/environment.py:882-898 (reset observation):
//   reset_place_kernel  one lane per env: Philox draws in the reference's draw order, rejection
//                       sampling of agents / goals, min_time from the PREVIOUS assignment (:545-547)
//   lexifair            fair goal assignment (fmarl_lexifair.hip; marl_fair_assign.py:16-55)
//   reset_emit_kernel   obs / node_obs / adj of the freshly reset envs (same emission code as step)
// The reference draws from NumPy's global MT19937; the device uses a counter-based Philox stream
// keyed (seed; draw, env, episode) so results do not depend on the number of GPUs.  oracle/philox.py
// is the bit-exact host twin used by the parity tests.
#pragma once
#include "fmarl_dev.h"
#include "fmarl_kernels.h"
#include "fmarl_step.hip"

namespace fmarl {

// Positions placed so far by one lane (= one env).  The rejection tests walk them sequentially; from
// global memory every test is a dependent ~1 us load, so an f32 copy is kept in LDS, slot-major
// [slot][lane] (8 B per lane, conflict-free).  A test is decided on the f32 copy unless the distance is
// within 1e-6 of the threshold, in which case the f64 value is re-read from global memory (exact result).
template <bool LDS>
struct Placed {
    float2 *lds;      // [slots][64]
    const Params &p;
    int env, lane;
    __device__ double2 g_obstacle(int k) const { return p.obstacle_pos[(size_t)env * p.O + k]; }
    __device__ double2 g_agent(int k) const { return p.agent_pos[(size_t)env * p.N + k]; }
    __device__ double2 g_landmark(int k) const { return p.landmark_pos[(size_t)env * p.L + k]; }
    __device__ void put(int slot, double2 x) { if (LDS) lds[slot * 64 + lane] = make_float2((float)x.x, (float)x.y); }
    __device__ void set_obstacle(int k, double2 x) { put(k, x); p.obstacle_pos[(size_t)env * p.O + k] = x; }
    __device__ void set_agent(int k, double2 x) { put(p.O + k, x); p.agent_pos[(size_t)env * p.N + k] = x; }
    __device__ void set_landmark(int k, double2 x) { put(p.O + p.N + k, x); p.landmark_pos[(size_t)env * p.L + k] = x; }
    // 0 = farther than thr, 1 = closer, 2 = too close to call in f32
    __device__ int coarse(int slot, float2 xf, float thr) const {
        const float2 q = lds[slot * 64 + lane];
        const float dx = q.x - xf.x, dy = q.y - xf.y, s = dx * dx + dy * dy;
        const float lo = (thr - 1e-6f) * (thr - 1e-6f), hi = (thr + 1e-6f) * (thr + 1e-6f);
        return s < lo ? 1 : (s > hi ? 0 : 2);
    }
    // any of the first k slots starting at `first` closer than thr to x (kind: 0 obstacle, 1 agent, 2 landmark)
    __device__ bool any_closer(int kind, int k, double2 x, double thr) const {
        bool hit = false;
        if (LDS) {
            const int first = kind == 0 ? 0 : (kind == 1 ? p.O : p.O + p.N);
            const float2 xf = make_float2((float)x.x, (float)x.y);
            bool unsure = false;
            for (int j = 0; j < k; ++j) {
                const int c = coarse(first + j, xf, (float)thr);
                hit |= c == 1;
                unsure |= c == 2;
            }
            if (!unsure) return hit;
            hit = false;
        }
        for (int j = 0; j < k; ++j)
            hit |= closer_than(kind == 0 ? g_obstacle(j) : (kind == 1 ? g_agent(j) : g_landmark(j)), x, thr);
        return hit;
    }
};

// navigation_graph.py:650-684 is_obstacle_collision(pos, size = 0.05)
template <class PL>
__device__ __forceinline__ bool obstacle_hit(const Params &p, const PL &pl, const double *wall, double2 x) {
    const bool plain = p.scenario == FMARL_SCENARIO_FORMATION;   // fair_graph_formation.py:518-530: no 1.05 factors
    const bool fairnav = p.scenario == FMARL_SCENARIO_FAIRNAV;   // nav_fairassign_...py:592-613: 2.0 (s+s), walls +-1.5 s
    bool hit = pl.any_closer(0, p.O, x, (fairnav ? 2.0 : 1.05) * (kEntitySize + kEntitySize));
    for (int w = 0; w < p.W; ++w) {
        const double axis = wall[w * 4], e0 = wall[w * 4 + 1], e1 = wall[w * 4 + 2];
        const int orient = (int)wall[w * 4 + 3];
        const double s = fairnav ? 1.5 * kEntitySize : kEntitySize / 2;
        const double pperp = orient == 0 ? x.y : x.x, ppar = orient == 0 ? x.x : x.y;
        hit |= (plain || fairnav) ? ((axis - s <= pperp) && (pperp <= axis + s) && (e0 - s <= ppar) && (ppar <= e1 + s))
                                  : wall_box_hit(x, axis, e0, e1, orient);
    }
    return hit;
}

// reset_world + random_scenario of ONE env by the calling lane (navigation_graph.py:212-575 and the two formation
// scenarios' variants): Philox draws in the reference's draw order, rejection sampling through `pl` (where the placed
// positions are kept for the tests), per-agent vectors reset.  Shared by reset_place_kernel and the in-kernel reset of
// fairnav_kernel (fmarl_fairnav.hip).  `stage`: positions only, into the staging fields `p` is bound to.
// `what`: kPlaceFull = everything; kPlaceStage = positions only, into the staging fields `p` is bound to; kPlaceEntities (the
// in-kernel reset of fairnav_pass) = the entities, walls, wall length, place_fails and the episode counter -- the per-agent vectors are
// reset by the env's own agent lanes in parallel, `episode_in` is the caller's copy of the episode counter and `pre` / `n_pre` the
// first blocks of the env's Philox stream drawn ahead by other lanes (PhiloxStream).
enum PlaceWhat { kPlaceFull = 0, kPlaceStage = 1, kPlaceEntities = 2 };
template <class PL>
__device__ __forceinline__ void place_env(const Params &p, PL &pl, int mode, int env, int what, int episode_in = 0,
                                          const double *pre = nullptr, int n_pre = 0) {
    const int N = p.N, L = p.L;
    const size_t a0 = (size_t)env * N;
    const bool stage = what != kPlaceFull;   // (the per-agent vectors are somebody else's)
    int episode = 0;
    if (mode == kResetInit) {
        for (int i = 0; i < N; ++i) { p.goal_match[a0 + i] = i; p.min_time[a0 + i] = __builtin_huge_val(); }
    } else {
        episode = what == kPlaceEntities ? episode_in : p.episode[env];
    }
    PhiloxStream rng(p.seed, (uint32_t)(p.env_offset + env), (uint32_t)episode, 0, pre, (uint32_t)n_pre);
    const bool formation = p.scenario == FMARL_SCENARIO_FORMATION, fairnav = p.scenario == FMARL_SCENARIO_FAIRNAV;
    if (mode == kResetInit)   // make_world: navigation_graph.py:183-185 (fairnav draws U(0.2, 0.4) there, then redraws)
        p.wall_length[env] = rng.uniform(0.2, fairnav ? 0.4 : 0.8) * p.world_size / 4;
    if (fairnav)              // nav_fairassign_...py:239-241: wall_length is drawn again at every reset_world
        p.wall_length[env] = rng.uniform(0.2, 0.8) * p.world_size / 4;
    const double ws = p.world_size, wlen = p.wall_length[env];
    const double goal_scale = formation ? 0.5 : 0.8;   // fair_graph_formation.py:363 vs navigation_graph.py:492

    if (!stage) {
        for (int i = 0; i < N; ++i) {   // :215-225, :239-240
            p.times_required[a0 + i] = -1.0; p.dists_to_goal[a0 + i] = -1.0; p.dist_left[a0 + i] = -1.0;
            p.num_obst_coll[a0 + i] = 0; p.num_agent_coll[a0 + i] = 0; p.p_dist[a0 + i] = 0.0;
        }
        p.cur_step[env] = 0;
    }
    for (int k = 0; k < p.O; ++k) {   // :271-275
        double2 u = rng.uniform_pair(-ws / 2, ws / 2);
        pl.set_obstacle(k, make_double2(0.8 * u.x, 0.8 * u.y));
    }
    const double wall_position = rng.uniform(0.2, 0.9);   // :288, drawn even without walls
    double wall[2 * 4] = {0, 0, 0, 0, 0, 0, 0, 0};   // W <= 2 (axis, e0, e1, orient)
    for (int w = 0; w < p.W; ++w) {   // :294-324
        size_t g = (size_t)env * p.W + w;
        const int orient = formation ? 1 : rng.choice_hv();   // fair_graph_formation.py:276: always 'V', no draw
        const double axis = (w == 0 ? wall_position : -wall_position) * ws / 2;
        p.wall_orient[g] = orient; p.wall_axis[g] = axis;
        if (what != kPlaceStage) { p.wall_e0[g] = -wlen; p.wall_e1[g] = wlen; }   // staged: derived from wall_length at commit
        if (w == 0) { wall[0] = axis; wall[1] = -wlen; wall[2] = wlen; wall[3] = orient; }
        else { wall[4] = axis; wall[5] = -wlen; wall[6] = wlen; wall[7] = orient; }
    }
    const double thr = 1.05 * (kEntitySize + kEntitySize);
    const double thr_goal = (fairnav ? 1.2 : 1.05) * (kEntitySize + kEntitySize);   // nav_fairassign_...py:643
    int fails = 0;   // placements accepted with every draw colliding (the reference would still be drawing)
    for (int k = 0, tries = 0; k < N;) {   // :389-457
        double2 x = rng.uniform_pair(-ws / 2, ws / 2);
        ++tries;
        bool bad = obstacle_hit(p, pl, wall, x);
        bad |= pl.any_closer(1, k, x, thr);   // :689-698
        if (!bad || tries >= kMaxTries) {
            fails += bad ? 1 : 0;
            pl.set_agent(k, x);
            if (!stage) p.agent_vel[a0 + k] = make_double2(0.0, 0.0);
            ++k; tries = 0;
        }
    }
    for (int k = 0, tries = 0; k < L;) {   // :472-535
        double2 u = rng.uniform_pair(-ws / 2, ws / 2);
        double2 x = make_double2(goal_scale * u.x, goal_scale * u.y);
        ++tries;
        bool bad = obstacle_hit(p, pl, wall, x);
        bad |= pl.any_closer(2, k, x, thr_goal);   // :707-716
        if (!bad || tries >= kMaxTries) { fails += bad ? 1 : 0; pl.set_landmark(k, x); ++k; tries = 0; }
    }
    p.place_fails[env] = fails;   // (staging: `p` is bound to the staging twin, committed with the rest)
    if (what == kPlaceEntities) p.episode[env] = episode + 1;
    if (stage) return;   // min_time, (staging:) episode counter: reset_commit_kernel
    if (formation) {
        // fair_graph_formation.py:394-417: slots on the circle about landmark 0, occupancy cleared;
        // min_time against the agent's OWN slot index (:573-580); formation_complete cleared (:231)
        const double2 L0 = pl.g_landmark(0);
        double tmin = 1e300;
        for (int i = 0; i < N; ++i) {
            const double2 a = pl.g_agent(i);
            double th = atan2(a.y - L0.y, a.x - L0.x);
            if (th < 0) th += 2 * M_PI;
            tmin = fmin(tmin, th);
        }
        for (int i = 0; i < N; ++i) {
            const double ang = tmin + i * ((2 * M_PI) / N);
            const double2 P = make_double2(L0.x + 0.5 * cos(ang), L0.y + 0.5 * sin(ang));
            p.slot_pos[a0 + i] = P; p.slot_occ[a0 + i] = 0.0; p.formation_done[a0 + i] = 0.0;
            if (p.has_max_speed) p.min_time[a0 + i] = dist2(pl.g_agent(i), P) / p.max_speed;
        }
    } else if (fairnav) {   // nav_fairassign_...py:233-238, :446-459: goal_match reset to arange BEFORE min_time
        for (int i = 0; i < N; ++i) {
            p.goal_occ[a0 + i] = 0.0; p.goal_history[a0 + i] = -1; p.goal_reached[a0 + i] = -1; p.status[a0 + i] = 0;
            if (p.has_max_speed) p.min_time[a0 + i] = dist2(pl.g_agent(i), pl.g_landmark(i)) / p.max_speed;
        }
    } else if (p.has_max_speed) {   // :545-547, :719-728 -- previous episode's goal_match_index
        for (int i = 0; i < N; ++i)
            p.min_time[a0 + i] = dist2(pl.g_agent(i), pl.g_landmark(p.goal_match[a0 + i])) / p.max_speed;
    }
    p.episode[env] = episode + 1;
}

template <bool LDS>
__global__ __launch_bounds__(64) void reset_place_kernel(Params p, int mode, const uint8_t *mask) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= p.n_envs) return;
    // kResetStage: the position / wall / goal_match pointers of `p` are bound to the staging fields and
    // only placement is done (for envs without valid staged data); everything else belongs to the commit.
    const bool stage = mode == kResetStage;
    bool doit = true;
    if (mode == kResetMask) doit = mask[env] != 0;
    else if (mode == kResetAuto) {   // all agents done (environment.py:237-247: status or episode length)
        doit = p.cur_step[env] >= p.episode_length;
        if (!doit && p.scenario == FMARL_SCENARIO_FAIRNAV) {
            doit = true;
            for (int i = 0; i < p.N; ++i) doit &= p.status[(size_t)env * p.N + i] != 0;
        }
    } else if (stage) doit = p.stage_valid[env] == 0;
    if (stage) p.stage_need[env] = doit ? 1 : 0;
    else p.reset_flag[env] = doit ? 1 : 0;
    if (!doit) return;
    if (!stage) p.stage_valid[env] = 0;   // a synchronous reset consumes this episode index: staged data is stale

    Placed<LDS> pl{(float2 *)lds_raw, p, env, (int)threadIdx.x};
    place_env(p, pl, mode, env, stage ? kPlaceStage : kPlaceFull);
}

// Asynchronous reset, step 2 of 2: make the staged episode the live one for the selected envs
// (navigation_graph.py:212-262 vector resets, :545-547 min_time against the PREVIOUS goal_match).
// One thread per (env, agent), same workgroup shape as the step kernel.
__global__ __launch_bounds__(kThreads) void reset_commit_kernel(Params p, int mode, const uint8_t *mask) {
    const int tid = threadIdx.x, N = p.N;
    const int env0 = env_block(p) * p.epb;
    const int nenv = min(p.epb, p.n_envs - env0);
    const int el = tid / N, i = tid - el * N;
    const bool active = el < nenv;
    const int env = env0 + el;
    bool doit = false;
    if (active) {
        doit = true;
        if (mode == kResetMask) doit = mask[env] != 0;
        else if (mode == kResetAuto) doit = p.cur_step[env] >= p.episode_length;
    }
    __syncthreads();   // every lane has read cur_step before lane 0 clears it
    if (!active) return;
    if (i == 0) p.reset_flag[env] = doit ? 1 : 0;
    if (!doit) return;
    const size_t g = (size_t)env * N + i;
    const double2 x = p.st_agent_pos[g];
    if (p.has_max_speed)
        p.min_time[g] = dist2(x, p.st_landmark_pos[(size_t)env * p.L + p.goal_match[g]]) / p.max_speed;
    p.agent_pos[g] = x; p.agent_vel[g] = make_double2(0.0, 0.0); p.p_dist[g] = 0.0;
    p.goal_match[g] = p.st_goal_match[g];
    p.times_required[g] = -1.0; p.dists_to_goal[g] = -1.0; p.dist_left[g] = -1.0;
    p.num_obst_coll[g] = 0; p.num_agent_coll[g] = 0;
    for (int k = i; k < p.L; k += N) p.landmark_pos[(size_t)env * p.L + k] = p.st_landmark_pos[(size_t)env * p.L + k];
    for (int k = i; k < p.O; k += N) p.obstacle_pos[(size_t)env * p.O + k] = p.st_obstacle_pos[(size_t)env * p.O + k];
    for (int w = i; w < p.W; w += N) {
        const size_t gw = (size_t)env * p.W + w;
        p.wall_axis[gw] = p.st_wall_axis[gw]; p.wall_orient[gw] = p.st_wall_orient[gw];
        p.wall_e0[gw] = -p.wall_length[env]; p.wall_e1[gw] = p.wall_length[env];
    }
    if (i == 0) { p.cur_step[env] = 0; p.episode[env] += 1; p.stage_valid[env] = 0; p.place_fails[env] = p.st_place_fails[env]; }
}

// Asynchronous reset, staging side: placement + assignment done -> the staged data is valid.
__global__ void stage_finish_kernel(Params p) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env < p.n_envs && p.stage_need[env]) p.stage_valid[env] = 1;
}

// obs / node_obs / adj of freshly reset envs (environment.py:882-898).  Same workgroup shape as the
// step kernel; envs whose reset_flag is 0 are skipped.
__global__ __launch_bounds__(kThreads) void reset_emit_kernel(Params p, FmarlOutputs o) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int env0 = env_block(p) * p.epb;
    const int nenv = min(p.epb, p.n_envs - env0);
    const int el = tid / p.N, i = tid - el * p.N;
    const bool active = el < nenv;
    const int env = env0 + el;
    const size_t g = (size_t)env * p.N + i;
    char *base = lds + (size_t)el * p.lds_env_bytes;
    double2 *s_pos = (double2 *)(base + p.lds_pos);
    double *s_stat = (double *)(lds + p.lds_stat + (size_t)el * p.stat_stride);
    double2 x = make_double2(0, 0), v = make_double2(0, 0);
    double Dg = 0, pdist = 0;
    bool flagged = false;
    if (active) {
        flagged = p.reset_flag[env] != 0;
        x = p.agent_pos[g]; v = p.agent_vel[g]; Dg = p.dists_to_goal[g];
        s_pos[i] = x;
        pdist = p.p_dist[g];
        if (!p.scan_stats) { s_stat[i] = pdist; s_stat[p.N + i] = Dg; }
        if (i == 0) *(int *)(base + p.lds_flag) = flagged ? 0 : 1;   // 1 = skip
    }
    if (!__syncthreads_or(flagged)) return;   // no freshly reset env in this workgroup (block-uniform exit)
    load_statics(p, lds, env0, nenv);
    double sm = 0.0, sq = 0.0;
    if (p.scan_stats) {   // every lane of the wave takes part (fmarl_dev.h); each agent then picks its vector
        double dm_, dq_;
        seg_all_runs(p.N, pdist, sm, sq);
        seg_all_runs(p.N, Dg, dm_, dq_);
        if (Dg != -1.0) { sm = dm_; sq = dq_; }
    }
    __syncthreads();
    if (active && flagged) {
        const double2 goal = s_pos[p.N + p.goal_match[g]];
        double m, sd;   // navigation_graph.py:849-854
        if (p.scan_stats) { m = sm; sd = sqrt(sq / p.N); }
        else if (Dg == -1.0) mixed_stats(s_stat, s_stat, p.N, p.N, m, sd);
        else mixed_stats(s_stat + p.N, s_stat + p.N, p.N, p.N, m, sd);
        const double fairness = ratio_out(m, sd + 0.0001);
        if (o.obs) {
            float *ob = o.obs + g * p.D;
            ob[0] = (float)v.x; ob[1] = (float)v.y; ob[2] = (float)x.x; ob[3] = (float)x.y;
            ob[4] = (float)(goal.x - x.x); ob[5] = (float)(goal.y - x.y); ob[6] = (float)fairness;
        }
        store_agent_rows(p, base, i, x, v, goal);
    }
    __syncthreads();
    emit_graph(p, o, lds, env0, nenv);
}

// navigation_graph.py:555 dist.cdist(agent_pos, goal_pos)
__global__ void cost_matrix_kernel(const double2 *agent_pos, const double2 *goal_pos, double *costs,
                                   int n_envs, int N, int L) {
    const size_t total = (size_t)n_envs * N * L;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
        size_t env = q / ((size_t)N * L);
        int r = (int)(q - env * N * L);
        int i = r / L, j = r - i * L;
        costs[q] = dist2(agent_pos[env * N + i], goal_pos[env * L + j]);
    }
}

}  // namespace fmarl
