// libfmarl.so -- C-ABI (include/fmarl.h) over the gfx950 kernels.  Single translation unit.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "fmarl_dev.h"
#include "fmarl_kernels.h"
#include "fmarl_step.hip"
#include "fmarl_reset.hip"
#include "fmarl_lexifair.hip"
#include "fmarl_formation.hip"
#include "fmarl_fairnav.hip"
#include "fmarl_graph.hip"
#include "fmarl_rebuild.hip"
#include "fmarl_learner.hip"

using namespace fmarl;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *a = "") {
    snprintf(g_err, sizeof g_err, fmt, a);
    return code;
}
#define HIP_OK(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return fail(FMARL_EHIP, #call ": %s", hipGetErrorString(e_)); \
    } while (0)

// node_obs rows of 16-byte multiples (navigation_graph with E * F % 4 == 0, the formation scenario) and adj with
// E % 4 == 0 are written with 16-byte stores straight from registers: those buffers must be 16-byte aligned
// (hipMalloc and torch allocations are).  Every other shape goes through aligned frames and takes any float pointer.
bool outputs_aligned(const fmarl::Params &p, const FmarlOutputs *o) {
    const bool form = p.scenario == FMARL_SCENARIO_FORMATION;
    if (o->node_obs && (p.vec_node || form) && ((uintptr_t)o->node_obs & 15)) return false;
    if (o->adj && p.vec_adj && ((uintptr_t)o->adj & 15)) return false;
    return true;
}

struct Layout {
    size_t off[FMARL_NUM_FIELDS], count[FMARL_NUM_FIELDS];
    int dtype[FMARL_NUM_FIELDS];
    size_t total;
};

bool config_ok(const FmarlConfig *c, const char **why) {
    *why = "";
    if (!c) { *why = "null config"; return false; }
    const bool form = c->scenario == FMARL_SCENARIO_FORMATION;
    const bool fnav = c->scenario == FMARL_SCENARIO_FAIRNAV;
    if (c->scenario != FMARL_SCENARIO_NAVIGATION_GRAPH && !form && !fnav) { *why = "unsupported scenario"; return false; }
    if (fnav && (c->num_agents < 2 || c->num_agents > 32)) { *why = "nav_fairassign_fairrew_formation_graph is built for num_agents in 2..32"; return false; }
    if (c->n_envs < 1) { *why = "n_envs < 1"; return false; }
    if (form && (c->num_agents < 1 || c->num_agents > 32)) { *why = "fair_graph_formation is built for num_agents in 1..32"; return false; }
    if (c->num_agents < 1 || c->num_agents > 64) { *why = "navigation_graph is built for num_agents in 1..64"; return false; }
    if (!form && c->num_landmarks != c->num_agents) { *why = "navigation_graph needs num_landmarks == num_agents"; return false; }
    if (form && c->num_landmarks < 1) { *why = "fair_graph_formation needs num_landmarks >= 1"; return false; }
    if (c->num_obstacles < 0 || c->num_obstacles > 4096) { *why = "bad num_obstacles"; return false; }
    if (c->num_walls < 0 || c->num_walls > 2) { *why = "num_walls must be 0..2"; return false; }
    if (form && c->num_walls != 2) { *why = "fair_graph_formation always has 2 walls"; return false; }
    if (c->episode_length < 1) { *why = "episode_length < 1"; return false; }
    if (c->envs_per_workgroup < 0 || c->reserved0 != 0) { *why = "envs_per_workgroup < 0 or reserved0 != 0 (zero-initialise FmarlConfig)"; return false; }
    if ((c->flags & FMARL_FLAG_GLOBAL_FEATURES) && (form || fnav || c->num_walls != 0)) {
        *why = "global node features: navigation_graph without walls only (the reference's _get_entity_feat_global knows no walls)";
        return false;
    }
    return true;
}

size_t dtype_bytes(int dt) { return dt == FMARL_DTYPE_F64 ? 8 : (dt == FMARL_DTYPE_I32 ? 4 : 1); }

void make_layout(const FmarlConfig *c, Layout *l) {
    const size_t n = c->n_envs, N = c->num_agents, L = c->num_landmarks, O = c->num_obstacles, W = c->num_walls;
    auto set = [&](int f, size_t count, int dt) { l->count[f] = count; l->dtype[f] = dt; };
    const bool form = c->scenario == FMARL_SCENARIO_FORMATION;
    set(FMARL_F_AGENT_POS, n * N * 2, FMARL_DTYPE_F64);   set(FMARL_F_AGENT_VEL, n * N * 2, FMARL_DTYPE_F64);
    set(FMARL_F_P_DIST, n * N, FMARL_DTYPE_F64);          set(FMARL_F_LANDMARK_POS, n * L * 2, FMARL_DTYPE_F64);
    set(FMARL_F_OBSTACLE_POS, n * O * 2, FMARL_DTYPE_F64); set(FMARL_F_WALL_AXIS, n * W, FMARL_DTYPE_F64);
    set(FMARL_F_WALL_E0, n * W, FMARL_DTYPE_F64);         set(FMARL_F_WALL_E1, n * W, FMARL_DTYPE_F64);
    set(FMARL_F_WALL_ORIENT, n * W, FMARL_DTYPE_I32);     set(FMARL_F_WALL_LENGTH, n, FMARL_DTYPE_F64);
    set(FMARL_F_GOAL_MATCH, n * N, FMARL_DTYPE_I32);      set(FMARL_F_DISTS_TO_GOAL, n * N, FMARL_DTYPE_F64);
    set(FMARL_F_TIMES_REQUIRED, n * N, FMARL_DTYPE_F64);  set(FMARL_F_DIST_LEFT, n * N, FMARL_DTYPE_F64);
    set(FMARL_F_NUM_OBST_COLL, n * N, FMARL_DTYPE_I32);   set(FMARL_F_NUM_AGENT_COLL, n * N, FMARL_DTYPE_I32);
    set(FMARL_F_MIN_TIME, n * N, FMARL_DTYPE_F64);        set(FMARL_F_CUR_STEP, n, FMARL_DTYPE_I32);
    set(FMARL_F_EPISODE, n, FMARL_DTYPE_I32);
    set(FMARL_F_SLOT_POS, form ? n * N * 2 : 0, FMARL_DTYPE_F64);
    set(FMARL_F_SLOT_OCC, form ? n * N : 0, FMARL_DTYPE_F64);
    set(FMARL_F_SLOT_DELTA, form ? n * N : 0, FMARL_DTYPE_F64);
    set(FMARL_F_FORMATION_DONE, form ? n * N : 0, FMARL_DTYPE_F64);
    const bool fnav = c->scenario == FMARL_SCENARIO_FAIRNAV;
    set(FMARL_F_GOAL_OCC, fnav ? n * N : 0, FMARL_DTYPE_F64);     set(FMARL_F_GOAL_HISTORY, fnav ? n * N : 0, FMARL_DTYPE_I8);
    set(FMARL_F_GOAL_REACHED, fnav ? n * N : 0, FMARL_DTYPE_I8); set(FMARL_F_STATUS, fnav ? n * N : 0, FMARL_DTYPE_I8);
    set(FMARL_F_RESET_FLAG, n, FMARL_DTYPE_I32);
    const bool async = (c->flags & FMARL_FLAG_ASYNC_RESET) && !form && !fnav;
    set(FMARL_F_STAGE_AGENT_POS, async ? n * N * 2 : 0, FMARL_DTYPE_F64);
    set(FMARL_F_STAGE_LANDMARK_POS, async ? n * L * 2 : 0, FMARL_DTYPE_F64);
    set(FMARL_F_STAGE_OBSTACLE_POS, async ? n * O * 2 : 0, FMARL_DTYPE_F64);
    set(FMARL_F_STAGE_WALL_AXIS, async ? n * W : 0, FMARL_DTYPE_F64);
    set(FMARL_F_STAGE_WALL_ORIENT, async ? n * W : 0, FMARL_DTYPE_I32);
    set(FMARL_F_STAGE_GOAL_MATCH, async ? n * N : 0, FMARL_DTYPE_I32);
    set(FMARL_F_STAGE_VALID, n, FMARL_DTYPE_I32);
    set(FMARL_F_STAGE_NEED, n, FMARL_DTYPE_I32);
    set(FMARL_F_PLACE_FAILS, n, FMARL_DTYPE_I32);
    set(FMARL_F_STAGE_PLACE_FAILS, async ? n : 0, FMARL_DTYPE_I32);
    set(FMARL_F_MATCH_DUAL, form ? n * N : 0, FMARL_DTYPE_F64);
    set(FMARL_F_ROT_TABLE, form ? 2 * N : 0, FMARL_DTYPE_F64);
    size_t off = 0;
    for (int f = 0; f < FMARL_NUM_FIELDS; ++f) {
        l->off[f] = off;
        size_t bytes = l->count[f] * dtype_bytes(l->dtype[f]);
        off += (bytes + 255) / 256 * 256;
    }
    l->total = off;
}

struct Handle {
    FmarlConfig cfg;
    Layout layout;
    Params base;        // everything but the state pointers
    size_t lds_bytes;
    int grid;
    int threads;        // workgroup size of the step / reset-emission launches (waves * 64)
    bool captured;      // a step of this handle was captured into a hipGraph: replays advance the device's step counters
                        // behind the host's back, so the host-side lockstep shortcut is off for good
    bool lockstep;      // all envs share one step counter, known on the host
    int host_step;
    bool episode_started;   // the last reset / step call may have started episodes (host-side knowledge, conservative)
    int device;         // HIP device the handle was created on (side stream, events, kernel attributes live there)
    bool async;         // FMARL_FLAG_ASYNC_RESET: next episode staged on `side`
    bool stage_dirty;   // staged data may be stale (caller wrote the state): next reset goes the synchronous way
    bool stage_pending; // a reset has consumed the staged episode; the next one is staged by the next fmarl_step call
    hipStream_t side;
    hipEvent_t ev_commit, ev_staged;
    unsigned long long cap_id;        // id of the stream capture the current call runs in (0 = not capturing)
    unsigned long long cap_stage_id;  // capture in which the staging of the next episode was last enqueued (0 = eagerly)
    size_t place_lds;   // dynamic LDS of reset_place_kernel<true> (0: positions stay in global memory)
    int span_threads;   // workgroup size of the span kernel (fairnav: 192 when the agent lanes fit three waves, else = threads)
    bool small_ok;      // navigation_graph: the small-batch kernels apply (fmarl_step.hip step_body SMALL) unless the call counts policy edges
    hipEvent_t *ev;     // profiling: 2 * ev_cap events around step-kernel launches
    int *ev_steps;      // env steps each profiled launch covers (a span launch: many)
    int ev_cap, ev_n;
    int64_t counts[4];  // fmarl_launch_counts
    double rot[64];     // formation: (cos, sin) of i * 2 pi / N
    double *d_rot;      // ... on the device, owned by the handle (512 bytes; the kernels read THIS table: a caller that zeroes,
                        // restores or copies its state buffer field by field cannot lose it -- ADVICE round 3).  The state
                        // buffer's FMARL_F_ROT_TABLE field keeps a copy for readers of older layouts; nothing reads it back
};

// Entry points that take a handle run on the handle's device whatever the caller's current device is
// (a handle created while cuda:1 was current keeps working from a thread whose current device is cuda:0).
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(const Handle *h) {
        int cur = -1;
        if (h && h->device >= 0 && hipGetDevice(&cur) == hipSuccess && cur != h->device && hipSetDevice(h->device) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int align16(int x) { return (x + 15) / 16 * 16; }

// Params.order for a launch of `grid` workgroups (fmarl_dev.h env_block): the odd number nearest the golden section of the grid
// that is coprime with it -- consecutive workgroups then work on env blocks far apart, and the blocks of any run of
// consecutive workgroups are spread evenly over the batch.  Small grids keep dispatch order (nothing to scatter).
int scatter_order(int grid) {
    if (grid < 64) return 1;
#ifdef FMARL_NO_SCATTER   // (A/B builds: tools/mkvariant.sh noscatter -DFMARL_NO_SCATTER)
    return 1;
#endif
#ifdef FMARL_MEASURE
    if (const char *e = getenv("FMARL_ORDER")) { const int v = atoi(e); if (v == 1) return 1; }
#endif
    int o = (int)(grid * 0.6180339887498949) | 1;
    for (;; o += 2) {
        int a = grid, b = o % grid;
        while (b) { const int r = a % b; a = b; b = r; }
        if (a == 1) return o;
    }
}

Params bind(const Handle *h, void *state) {
    Params p = h->base;
    char *s = (char *)state;
    const size_t *o = h->layout.off;
    p.agent_pos = (double2 *)(s + o[FMARL_F_AGENT_POS]);       p.agent_vel = (double2 *)(s + o[FMARL_F_AGENT_VEL]);
    p.landmark_pos = (double2 *)(s + o[FMARL_F_LANDMARK_POS]); p.obstacle_pos = (double2 *)(s + o[FMARL_F_OBSTACLE_POS]);
    p.p_dist = (double *)(s + o[FMARL_F_P_DIST]);              p.wall_axis = (double *)(s + o[FMARL_F_WALL_AXIS]);
    p.wall_e0 = (double *)(s + o[FMARL_F_WALL_E0]);            p.wall_e1 = (double *)(s + o[FMARL_F_WALL_E1]);
    p.wall_length = (double *)(s + o[FMARL_F_WALL_LENGTH]);    p.dists_to_goal = (double *)(s + o[FMARL_F_DISTS_TO_GOAL]);
    p.times_required = (double *)(s + o[FMARL_F_TIMES_REQUIRED]); p.dist_left = (double *)(s + o[FMARL_F_DIST_LEFT]);
    p.min_time = (double *)(s + o[FMARL_F_MIN_TIME]);          p.wall_orient = (int *)(s + o[FMARL_F_WALL_ORIENT]);
    p.goal_match = (int *)(s + o[FMARL_F_GOAL_MATCH]);         p.num_obst_coll = (int *)(s + o[FMARL_F_NUM_OBST_COLL]);
    p.num_agent_coll = (int *)(s + o[FMARL_F_NUM_AGENT_COLL]); p.cur_step = (int *)(s + o[FMARL_F_CUR_STEP]);
    p.episode = (int *)(s + o[FMARL_F_EPISODE]);               p.reset_flag = (int *)(s + o[FMARL_F_RESET_FLAG]);
    p.slot_pos = (double2 *)(s + o[FMARL_F_SLOT_POS]);         p.slot_occ = (double *)(s + o[FMARL_F_SLOT_OCC]);
    p.slot_delta = (double *)(s + o[FMARL_F_SLOT_DELTA]);      p.formation_done = (double *)(s + o[FMARL_F_FORMATION_DONE]);
    p.goal_occ = (double *)(s + o[FMARL_F_GOAL_OCC]);          p.goal_history = (int8_t *)(s + o[FMARL_F_GOAL_HISTORY]);
    p.goal_reached = (int8_t *)(s + o[FMARL_F_GOAL_REACHED]);  p.status = (int8_t *)(s + o[FMARL_F_STATUS]);
    p.st_agent_pos = (double2 *)(s + o[FMARL_F_STAGE_AGENT_POS]);   p.st_landmark_pos = (double2 *)(s + o[FMARL_F_STAGE_LANDMARK_POS]);
    p.st_obstacle_pos = (double2 *)(s + o[FMARL_F_STAGE_OBSTACLE_POS]); p.st_wall_axis = (double *)(s + o[FMARL_F_STAGE_WALL_AXIS]);
    p.st_wall_orient = (int *)(s + o[FMARL_F_STAGE_WALL_ORIENT]);   p.st_goal_match = (int *)(s + o[FMARL_F_STAGE_GOAL_MATCH]);
    p.stage_valid = (int *)(s + o[FMARL_F_STAGE_VALID]);            p.stage_need = (int *)(s + o[FMARL_F_STAGE_NEED]);
    p.match_dual = (double *)(s + o[FMARL_F_MATCH_DUAL]);
    p.rot_table = h->d_rot ? (const double2 *)h->d_rot : (const double2 *)(s + o[FMARL_F_ROT_TABLE]);
    p.place_fails = (int *)(s + o[FMARL_F_PLACE_FAILS]);             p.st_place_fails = (int *)(s + o[FMARL_F_STAGE_PLACE_FAILS]);
    return p;
}

void launch_place(Handle *h, const Params &p, int mode, const uint8_t *mask, hipStream_t st) {
    const int blocks = (p.n_envs + 63) / 64;
    if (h->place_lds)
        hipLaunchKernelGGL(reset_place_kernel<true>, dim3(blocks), dim3(64), h->place_lds, st, p, mode, mask);
    else
        hipLaunchKernelGGL(reset_place_kernel<false>, dim3(blocks), dim3(64), 0, st, p, mode, mask);
}

// Stage the next episode of every env without valid staged data: placement + fair assignment into the
// staging fields, on the side stream, ordered after everything `st` has done so far.
//
// Inside a stream capture (FMARL_RESET_LOCKSTEP graphs) the same calls make the side stream a forked branch of the caller's
// graph: the event recorded on the capturing stream pulls `side` into the capture, the staging kernels become nodes beside
// the step kernels of the episode, and the episode-ending launch joins the branch through ev_staged.  A graph must not rely
// on what the host knew about the staged data when it was captured, so a captured staging always clears the validity flags
// first and leaves the host's own `stage_dirty` as it was (nothing ran).
// nav_fairassign_fairrew_formation_graph kernels: the instantiation for the handle's workgroup size (TH = 192 or 256 threads)
// navigation_graph: the step kernels' instances with the shape as compile-time constants (fmarl_step.hip kNavShapes; 0 = the generic ones).
// The small-batch kernels exist for shape 1 (BASELINE config 2), the full-batch ones for shape 2 (the reference's own 10-agent scale);
// FMARL_GENERIC_SHAPES=1 in the environment switches every scenario's shape instances off (A/B, escape hatch).
static bool generic_shapes() { static const bool g = [] { const char *e = getenv("FMARL_GENERIC_SHAPES"); return e && e[0] == '1'; }(); return g; }
static inline int nav_shape(const FmarlConfig *c) {
    if (generic_shapes() || c->scenario != FMARL_SCENARIO_NAVIGATION_GRAPH) return 0;
    for (int k = 1; k < (int)(sizeof(fmarl::kNavShapes) / sizeof(fmarl::kNavShapes[0])); ++k)
        if (c->num_agents == fmarl::kNavShapes[k].N && c->num_landmarks == c->num_agents && c->num_obstacles == fmarl::kNavShapes[k].O &&
            c->num_walls == fmarl::kNavShapes[k].W) return k;
    return 0;
}
#define FMARL_NAV_SMALL(h, kernel, grid, threads, lds, st, ...)                                                         \
    do {                                                                                                                \
        if (nav_shape(&(h)->cfg) == 1) { constexpr int SH = 1; hipLaunchKernelGGL(kernel, grid, threads, lds, st, __VA_ARGS__); } \
        else { constexpr int SH = 0; hipLaunchKernelGGL(kernel, grid, threads, lds, st, __VA_ARGS__); }                    \
    } while (0)
#define FMARL_NAV_FULL(h, kernel, grid, threads, lds, st, ...)                                                          \
    do {                                                                                                                \
        if (nav_shape(&(h)->cfg) == 2) { constexpr int SH = 2; hipLaunchKernelGGL(kernel, grid, threads, lds, st, __VA_ARGS__); } \
        else { constexpr int SH = 0; hipLaunchKernelGGL(kernel, grid, threads, lds, st, __VA_ARGS__); }                    \
    } while (0)
// (the launch that ends an episode, step_end_kernel: also for shape 1 -- a small batch's spans end in it once per episode, and at 4 096 x 3
// it is a fifth of the episode's kernel time)
#define FMARL_NAV_END(h, kernel, grid, threads, lds, st, ...)                                                           \
    do {                                                                                                                \
        const int sh_ = nav_shape(&(h)->cfg);                                                                           \
        if (sh_ == 1) { constexpr int SH = 1; hipLaunchKernelGGL(kernel, grid, threads, lds, st, __VA_ARGS__); }           \
        else if (sh_ == 2) { constexpr int SH = 2; hipLaunchKernelGGL(kernel, grid, threads, lds, st, __VA_ARGS__); }      \
        else { constexpr int SH = 0; hipLaunchKernelGGL(kernel, grid, threads, lds, st, __VA_ARGS__); }                    \
    } while (0)
// fair_graph_formation in BASELINE config 4's shape (fmarl_formation.hip formation_shape_const)
static inline bool form_shape1(const FmarlConfig *c) { return !generic_shapes() && c->num_agents == 10 && c->num_landmarks == 1 && c->num_obstacles == 3 && c->num_walls == 2; }
#define FMARL_FORM(h, kernel, grid, threads, lds, st, ...)                                                              \
    do {                                                                                                                \
        if (form_shape1(&(h)->cfg)) { constexpr int SH = 1; hipLaunchKernelGGL(kernel, grid, threads, lds, st, __VA_ARGS__); }    \
        else { constexpr int SH = 0; hipLaunchKernelGGL(kernel, grid, threads, lds, st, __VA_ARGS__); }                    \
    } while (0)
// nav_fairassign_fairrew_formation_graph in the shape of the shipped FA / FA+FR weights (3 agents, 3 goals, 3 obstacles, no wall): the step
// kernels' NL = 3 instances hold these counts as compile-time constants
// (... and at the size BASELINE.md section 2 times the reference at: 10 agents, 10 goals, 3 obstacles -- the per-step kernel only: beyond five
// agents fmarl_step_span launches per step)
static inline bool fnav_shape10(const FmarlConfig *c) { return !generic_shapes() && c->num_agents == 10 && c->num_landmarks == 10 && c->num_obstacles == 3 && c->num_walls == 0; }
static inline bool fnav_shape3(const FmarlConfig *c) { return !generic_shapes() && c->num_agents == 3 && c->num_landmarks == 3 && c->num_obstacles == 3 && c->num_walls == 0; }
// (the per-step, reset-observation and rebuild kernels always run kThreads wide: only the span kernel has a three-wave form)
#define FMARL_FNAV(h, kernel, grid, lds, st, ...) do { constexpr int TH = kThreads; hipLaunchKernelGGL(kernel, grid, dim3(TH), lds, st, __VA_ARGS__); } while (0)
// the step kernels also exist with the agent / goal count as a compile-time constant (NL = 3: the shipped FA / FA+FR configuration; 0 = any)
#define FMARL_FNAV_NL(h, kernel, grid, lds, st, ...)                                                    \
    do {                                                                                                \
        constexpr int TH = kThreads;                                                                    \
        if (fnav_shape3(&(h)->cfg)) { constexpr int NL = 3; hipLaunchKernelGGL(kernel, grid, dim3(TH), lds, st, __VA_ARGS__); } \
        else if (fnav_shape10(&(h)->cfg)) { constexpr int NL = 10; hipLaunchKernelGGL(kernel, grid, dim3(TH), lds, st, __VA_ARGS__); } \
        else { constexpr int NL = 0; hipLaunchKernelGGL(kernel, grid, dim3(TH), lds, st, __VA_ARGS__); }  \
    } while (0)
#define FMARL_FNAV_T(h, threads, kernel, grid, lds, st, ...)                                            \
    do {                                                                                                \
        const bool nl3_ = fnav_shape3(&(h)->cfg);                                                                         \
        if ((threads) == 192 && nl3_) { constexpr int TH = 192, NL = 3; hipLaunchKernelGGL(kernel, grid, dim3(TH), lds, st, __VA_ARGS__); } \
        else if ((threads) == 192) { constexpr int TH = 192, NL = 0; hipLaunchKernelGGL(kernel, grid, dim3(TH), lds, st, __VA_ARGS__); } \
        else if (nl3_) { constexpr int TH = 256, NL = 3; hipLaunchKernelGGL(kernel, grid, dim3(TH), lds, st, __VA_ARGS__); }             \
        else { constexpr int TH = 256, NL = 0; hipLaunchKernelGGL(kernel, grid, dim3(TH), lds, st, __VA_ARGS__); }                     \
    } while (0)

int launch_stage(Handle *h, void *state, hipStream_t st) {
    HIP_OK(hipEventRecord(h->ev_commit, st));
    HIP_OK(hipStreamWaitEvent(h->side, h->ev_commit, 0));
    Params p = bind(h, state);
    if (h->stage_dirty || h->cap_id) HIP_OK(hipMemsetAsync(p.stage_valid, 0, (size_t)p.n_envs * sizeof(int), h->side));
    if (!h->cap_id) h->stage_dirty = false;
    h->cap_stage_id = h->cap_id;
    Params q = p;   // same kernels, pointers bound to the staging fields
    q.agent_pos = p.st_agent_pos; q.landmark_pos = p.st_landmark_pos; q.obstacle_pos = p.st_obstacle_pos;
    q.wall_axis = p.st_wall_axis; q.wall_orient = p.st_wall_orient; q.goal_match = p.st_goal_match;
    q.reset_flag = p.stage_need; q.place_fails = p.st_place_fails;
    launch_place(h, q, kResetStage, nullptr, h->side);
    launch_lexifair_state(q, h->side);
    hipLaunchKernelGGL(stage_finish_kernel, dim3((p.n_envs + 255) / 256), dim3(256), 0, h->side, p);
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(h->ev_staged, h->side));
    ++h->counts[3];
    return FMARL_OK;
}

int launch_reset(Handle *h, void *state, int mode, const uint8_t *mask, const FmarlOutputs *outs, hipStream_t st) {
    Params p = bind(h, state);
    const bool form = p.scenario == FMARL_SCENARIO_FORMATION;
    if (h->stage_pending) {   // two resets without a step in between: stage now, the commit below then waits for it
        h->stage_pending = false;
        int rc = launch_stage(h, state, st);
        if (rc) return rc;
    }
    if (h->async && h->cap_id && h->cap_stage_id != h->cap_id)
        return fail(FMARL_EINVAL, "reset inside a stream capture: the staging of this episode is not part of the capture "
                                  "(capture whole episodes, starting with the first step after a reset)");
    const bool staged = h->async && (h->cap_id ? true : !h->stage_dirty) && mode != kResetInit;
    if (h->async) HIP_OK(hipStreamWaitEvent(st, h->ev_staged, 0));   // staging in flight must finish first
    if (staged) {   // commit the episode staged on the side stream
        hipLaunchKernelGGL(reset_commit_kernel, dim3(h->grid), dim3(kThreads), 0, st, p, mode, mask);
    } else {
        launch_place(h, p, mode, mask, st);
        if (!form) launch_lexifair_state(p, st);
    }
    if (outs && (outs->obs || outs->node_obs || outs->adj)) {
        if (form)
            hipLaunchKernelGGL((formation_kernel<false, 0>), dim3(h->grid), dim3(h->threads), h->lds_bytes, st, p, *outs,
                               (const int32_t *)nullptr, (const float *)nullptr, 0);
        else if (p.scenario == FMARL_SCENARIO_FAIRNAV)
            FMARL_FNAV(h, (fairnav_kernel<false, TH, 0>), dim3(h->grid), h->lds_bytes, st, p, *outs, (const int32_t *)nullptr, (const float *)nullptr, 0);
        else
            hipLaunchKernelGGL(reset_emit_kernel, dim3(h->grid), dim3(kThreads), h->lds_bytes, st, p, *outs);
    } else if ((form || p.scenario == FMARL_SCENARIO_FAIRNAV) && mode != kResetInit) {
        return fail(FMARL_EINVAL, "fmarl_reset: this scenario needs output buffers (its reset observation updates scenario state)");
    }
    HIP_OK(hipGetLastError());
    // The next episode is staged by the NEXT fmarl_step call, not here: work enqueued now would sit behind the caller's
    // "this step is done" synchronisation although it belongs to an episode that has not begun (a timed region that ends
    // on an episode boundary would pay milliseconds of placement + assignment for steps it never ran).
    if (h->async) h->stage_pending = true;
    return FMARL_OK;
}

}  // namespace

extern "C" {

const char *fmarl_last_error(void) { return g_err; }
int fmarl_state_changed(void *handle);

size_t fmarl_state_bytes(const FmarlConfig *cfg) {
    const char *why;
    if (!config_ok(cfg, &why)) { fail(FMARL_EINVAL, "fmarl_state_bytes: %s", why); return 0; }
    Layout l;
    make_layout(cfg, &l);
    return l.total;
}

int fmarl_state_field(const FmarlConfig *cfg, int field, size_t *offset_bytes, size_t *count, int *dtype) {
    const char *why;
    if (!config_ok(cfg, &why)) return fail(FMARL_EINVAL, "fmarl_state_field: %s", why);
    if (field < 0 || field >= FMARL_NUM_FIELDS) return fail(FMARL_EINVAL, "fmarl_state_field: bad field id");
    Layout l;
    make_layout(cfg, &l);
    if (offset_bytes) *offset_bytes = l.off[field];
    if (count) *count = l.count[field];
    if (dtype) *dtype = l.dtype[field];
    return FMARL_OK;
}

int fmarl_create(const FmarlConfig *cfg, void **handle) {
    const char *why;
    if (!handle) return fail(FMARL_EINVAL, "fmarl_create: null handle pointer");
    if (!config_ok(cfg, &why)) return fail(FMARL_EINVAL, "fmarl_create: %s", why);
    Handle *h = new (std::nothrow) Handle();
    if (!h) return fail(FMARL_EINVAL, "fmarl_create: out of host memory");
    h->cfg = *cfg;
    if (hipGetDevice(&h->device) != hipSuccess) h->device = -1;   // no GPU here: the handle still describes the layout
    make_layout(cfg, &h->layout);
    Params &p = h->base;
    memset(&p, 0, sizeof p);
    p.n_envs = cfg->n_envs; p.N = cfg->num_agents; p.L = cfg->num_landmarks; p.O = cfg->num_obstacles;
    const bool form = cfg->scenario == FMARL_SCENARIO_FORMATION;
    const bool fnav = cfg->scenario == FMARL_SCENARIO_FAIRNAV;
    p.W = cfg->num_walls; p.E = p.N + p.L + p.O + p.W; p.D = form ? 6 : (fnav ? 11 : 7); p.F = form ? 12 : (fnav ? 13 : 11);
    p.feat_global = (cfg->flags & FMARL_FLAG_GLOBAL_FEATURES) ? 1 : 0;
    if (p.feat_global) p.F = 7;
    p.min_obs_dist = cfg->min_obs_dist;
    p.episode_length = cfg->episode_length; p.has_max_speed = cfg->has_max_speed; p.env_offset = cfg->env_offset;
    p.scenario = cfg->scenario;
    p.world_size = cfg->world_size; p.max_speed = cfg->max_speed; p.collision_rew = cfg->collision_rew;
    p.goal_rew = cfg->goal_rew; p.thr = cfg->min_dist_thresh; p.fair_rew = cfg->fair_rew; p.zeroshift = cfg->zeroshift;
    p.seed = cfg->seed;
    p.edge_thr = (float)cfg->max_edge_dist;
    // per-env LDS layout (fmarl_step.hip EnvLds)
    int off = 0;
    p.lds_pos = off;    off = align16(off + p.E * 16);
    p.lds_agentf = off; off = align16(off + p.N * (form ? 8 : 16));   // formation: (vx, vy) only
    p.lds_ego = off;    off = align16(off + ((form || fnav) ? 0 : p.N * kEgoWidth * 4));
    p.scan_stats = !form && !fnav && p.N <= 64 && (p.N & (p.N - 1)) == 0;
    // generic row widths (everything but navigation_graph with E * F % 4 == 0 and the formation scenario's 48-byte
    // rows) leave through one LDS window per wave: fmarl_step.hip flush_rows
    const bool staged = fnav || (!form && (p.E * p.F) % 4 != 0) || (!form && !fnav && p.E * p.F / 4 > 64 * 4);
    // navigation_graph: the statistics blocks are dead once the emission starts (a workgroup barrier apart), so they share
    // the region of the emission windows instead of sitting in every env's table (one more workgroup per CU at N = 10)
    const bool stat_shared = staged && !form && !fnav && !p.scan_stats;
    p.lds_stat = off;   off = align16(off + ((p.scan_stats || stat_shared || form || fnav) ? 0 : 5 * p.N * 8));   // wave scans need no table; formation / fairnav: below
    p.lds_wall = off;   off = align16(off + p.W * 4 * 8);
    p.lds_flag = off;   off = align16(off + 16);   // flag, then the formation scenario's three occupancy words or the env's policy-edge counter
    p.lds_cnt = p.lds_flag + 4;   // (formation: the occupancy word of the previous pass, dead once the emission starts)
    p.has_posf = 1;   // f32 copy of the entity positions: adj and the node rows start from it (all three scenarios)
    p.lds_posf = off;   off = align16(off + (p.has_posf ? p.E * 8 : 0));
    p.has_wallf = 1;   // f32 wall corner words (round 5: the formation scenario too -- its kernels run four workgroups per CU by registers, and at 24 envs per workgroup the LDS has 300 bytes per env to spare: the rows no longer convert corners and slots per row)
    p.lds_wallf = off;  off = align16(off + (p.has_wallf ? p.W * 16 : 0));
    p.lds_constf = off; off = align16(off + (!form && !fnav ? 16 : 0));
    int form_dead = 0;   // bytes per env in the second LDS region (formation, fairnav)
    if (form) {   // fmarl_formation.hip FormLds
        p.f_slot_new = off; off = align16(off + p.N * 16);
        p.f_slotf = off;    off = align16(off + p.N * 8);
        p.f_g = off;        off = align16(off + 3 * p.N + 1);
        p.f_masks = off;    off = align16(off + (3 * p.N * 4 > p.N * 8 ? 3 * p.N * 4 : p.N * 8));
        // Tables nobody reads once the emission starts live in a region of their own behind all envs' blocks (offsets relative
        // to the env's part of it, f_dead_bytes each): the envs of one wave are consecutive there, and their part doubles as
        // the wave's emission window (rows leave as contiguous 16-byte chunks instead of 16 bytes per lane at a 48-byte stride)
        int d = 0;
        p.lds_stat = d;   d = align16(d + 2 * p.N * 8 + 12);   // [pd | Dg_old] + the mask of agents still under way
        p.f_slot_old = d; d = align16(d + p.N * 16);
        p.f_theta = d;    d = align16(d + p.N * 8);
        p.f_words = d;    d = align16(d + (p.N * 8 > (p.N + 2) * 4 ? p.N * 8 : (p.N + 2) * 4));   // column potentials of the last matching, then the walk's entity sets
        form_dead = p.lds2_bytes = d;
    }
    if (fnav) {   // fmarl_fairnav.hip FairNavLds
        // the env's block packed to 8 bytes, 8-byte row records: 352 + 272 = 624 bytes per env at N = 3, i.e. 64 envs per
        // workgroup in 39 KB = 1 024 workgroups for 65 536 envs, ALL resident at once (four per CU).  With 640 bytes per env the
        // workgroup took exactly 40 960 bytes and only three fit a CU: 768 workgroups ran, then the other 256 -- two generations
        // of a 31 us chain, 65 us per launch; 40 320 bytes is the most that was seen to fit four (profiles/archive/r3_notes.md)
        int o2 = 0;
        p.lds_pos = o2;    o2 += p.E * 16;
        p.lds_agentf = o2; o2 += p.N * 16;
        p.lds_wall = o2;   o2 += p.W * 32;
        p.lds_wallf = o2;  o2 += p.W * 16;
        p.lds_posf = o2;   o2 += p.E * 8;      // (16-byte reads of it only when E % 4 == 0: the table then is a multiple of 32 bytes)
        p.n_rows = o2;     o2 += p.N * p.N * 8;
        p.lds_flag = o2;   o2 += 12;   // flag, policy-edge counter, the env's episode counter (FairNavLds::episode)
        p.lds_cnt = p.lds_flag + 4;
        p.lds_constf = p.lds_ego = o2;
        off = align16(o2);
        // Tables nobody reads once the emission starts: a region of their own behind all envs' blocks, which the waves' emission
        // windows alias (13.5 KB that used to sit beside the envs' tables: 36 -> 58 envs per workgroup in the shipped FA+FR
        // configuration, and the launch time falls with the number of workgroups: an envs-per-workgroup sweep of round 3, tools/archive/epb_sweep.py in the history up to 28d23e2)
        int d = 0;
        p.lds_stat = d;    d += 5 * p.N * 8;        // [pd_new | Dg_old | Dg_new | Tr_old | Tr_new] x N
        p.n_D = d;         d += p.N * p.L * 8;
        p.n_minprox = d;   d += p.L * 8;
        p.n_occ = d;       d += p.L * 8 + (p.L + 7) / 8 * 8;   // occupancy (float64: fractional values occur), history (bytes)
        p.n_match = d;     d += p.N * 4;
        p.n_words = d;     d += 12;
        form_dead = p.lds2_bytes = (d + 7) / 8 * 8;
        // in-kernel reset: this many blocks of the new episode's Philox stream are drawn side by side into the bytes below n_words
        // (wall length, obstacles, wall position, wall orientations, agents, goals + a few rejected draws)
        p.n_pre = p.n_words / 16 < p.O + p.W + p.N + p.L + 6 ? p.n_words / 16 : p.O + p.W + p.N + p.L + 6;
    }
    p.lds_env_bytes = off;
    p.stage_wave_bytes = staged ? align16(kStageRows * p.F * 4 + 64) : 0;   // + the window's offset inside its 64-byte aligned frame
    int epb = kThreads / p.N;
    int shared_bytes = (kThreads / 64) * p.stage_wave_bytes;   // the windows' region (also holds the shared statistics blocks)
    if (stat_shared && epb * 5 * p.N * 8 > shared_bytes) shared_bytes = epb * 5 * p.N * 8;
    // about 40 KB per workgroup = four workgroups per CU (160 KB; kFourPerCu below: not the full quarter).  With 48 KB the shipped nav_fairassign configuration took 47 envs
    // per workgroup at three per CU: 0.082 ms per launch against 0.078 with 36 envs at four per CU (tools/epb_probe.sh fnav);
    // navigation_graph at 10 agents had gained 7 % from the same 3 -> 4 step.
    // fairnav: the windows alias the second region (the tables there are dead by then): the region holds whichever is larger
    const int kFourPerCu = 40320;   // LDS of a workgroup such that four share a CU: 40 960 = 160 KB / 4 does NOT fit four (measured)
    const int budget = kFourPerCu - (fnav ? 0 : shared_bytes);
    const int env_lds = p.lds_env_bytes + form_dead;   // LDS of one env incl. its share of the second region (formation, fairnav)
    if (epb * env_lds > budget) epb = budget / env_lds;
    if (fnav) while (epb > 1 && epb * p.lds_env_bytes + (epb * form_dead > shared_bytes ? epb * form_dead : shared_bytes) > kFourPerCu) --epb;
    if (epb < 1) epb = 1;
    // fair_graph_formation: every env lives inside one wave (fmarl_formation.hip), 64 / N envs per wave, and no wave ever waits
    // for another one.  Workgroups of ONE wave (the scheduler placing 64-lane units) measured slower than four waves per
    // workgroup: 0.295 vs 0.274 ms per launch at BASELINE config 4, two waves 0.276-0.286 (profiles/archive/r3_cfg4_notes.md)
    int form_waves = kThreads / 64;
#ifdef FMARL_MEASURE
    if (const char *e = getenv("FMARL_FORM_WAVES")) form_waves = atoi(e) >= 1 && atoi(e) <= 4 ? atoi(e) : form_waves;
#endif
    h->threads = form ? 64 * form_waves : kThreads;   // (fairnav: decided below, once the envs per workgroup are known)
    if (form) {
        int epw = 64 / p.N;
        if (epw < 1) { delete h; return fail(FMARL_EINVAL, "fmarl_create: fair_graph_formation is built for num_agents <= 32"); }
        while (epw > 1 && form_waves * epw * env_lds > budget) --epw;
        p.epw = epw;
        epb = form_waves * epw;
    }
    if (cfg->envs_per_workgroup > 0) {   // the caller's geometry (parity tests at the full-batch shape on a few envs)
        if (cfg->envs_per_workgroup < epb) epb = cfg->envs_per_workgroup;
    } else if ((p.n_envs + epb - 1) / epb < 512) {
        // small batches: spread over the 256 CUs (>= 512 workgroups) rather than fill every lane of a few
        int e2 = p.n_envs / 512; epb = e2 < 1 ? 1 : (e2 < epb ? e2 : epb);
    }
    if (!form && !fnav && cfg->envs_per_workgroup == 0 && (p.n_envs + epb - 1) / epb >= 512) {
        // navigation_graph rows of generic shape (E F not a multiple of 4): a workgroup's node rows leave as windows of 64 rows -- 2 816 bytes
        // = 44 lines at F = 11 -- so when the workgroup's region starts on a 64-byte boundary, every window does.  The store pattern alone
        // then runs 13 % faster (fmarl_store_pattern, 10 agents: 0.158 -> 0.137 ms per step), the kernel 3-5 % (spans 0.195 -> 0.189, one
        // launch per step 0.201 -> 0.191 at 24 instead of 25 envs per workgroup: profiles/r6_n10_tcc.md).  The largest env count whose
        // node bytes are a multiple of 64, if it gives up at most a tenth of the lanes.
        const long long env_node = 4LL * p.N * p.E * p.F;
        if ((p.E * p.F) % 4 != 0) {
            long long a = 64, b = env_node % 64;
            while (b) { const long long r = a % b; a = b; b = r; }
            const int g = (int)(64 / a), aligned = epb / g * g;
            if (aligned >= 1 && aligned * 10 >= epb * 9) epb = aligned;
        }
    }
#ifdef FMARL_MEASURE
    if (const char *e = getenv("FMARL_EPB")) { const int v = atoi(e); if (v >= 1 && v <= epb) epb = v; }   // envs per workgroup (experiments)
#endif
    if (form) { p.epw = (epb + form_waves - 1) / form_waves; epb = p.epw * form_waves; }
#ifdef FMARL_MEASURE
    // experiments: fewer envs in the LAST wave of a formation workgroup (the envs' blocks shrink, the waves' windows do not)
    if (const char *e = getenv("FMARL_FORM_EPB")) { const int v = atoi(e); if (form && v > (form_waves - 1) * p.epw && v <= epb) epb = v; }
#endif
    if ((size_t)env_lds > 160 * 1024) { delete h; return fail(FMARL_EINVAL, "fmarl_create: one env does not fit LDS"); }
    p.epb = epb;
    // nav_fairassign_fairrew_formation_graph: three waves when the envs' agent lanes fit them (64 envs x 3 agents: a fourth wave would
    // hold no agent) -- 168 vector registers per lane at four workgroups per CU instead of 128 (fmarl_fairnav.hip FairnavCarry)
    // -- for the span kernel only, which carries the state: one launch per step measured 4 % slower on three waves (the emission is
    // shared by fewer waves) and has no carry to fit
    // (only where the third wave is at least half full: with two waves of agents or little more -- 13 envs x 10 agents -- the three-wave form measured
    // 23 % SLOWER than one launch per step, 0.776 against 0.631 ms per step, every phase of the step alike: profiles/r6_fnav_spans_by_n.txt)
    h->span_threads = (fnav && epb * p.N <= 192 && epb * p.N >= 160) ? 192 : h->threads;
    h->small_ok = false;   // (decided below, once the emission shapes are known)
    p.lds_stage = align16(epb * p.lds_env_bytes);
    if (form) {   // the second region: epw envs per wave, the wave's part = its emission window
        p.stage_wave_bytes = p.epw * form_dead;
        const int rows = (p.stage_wave_bytes / 16 - 3) / 3;   // 48-byte rows + up to 3 chunks of alignment (formation_flush_rows)
        p.f_rows = rows > 63 ? 63 : rows;
    }
    size_t stage_bytes = (size_t)(h->threads / 64) * p.stage_wave_bytes;
    if (fnav && (size_t)epb * form_dead > stage_bytes) stage_bytes = (size_t)epb * form_dead;
    p.stat_stride = p.lds_env_bytes;
    if (stat_shared) {
        p.lds_stat = p.lds_stage; p.stat_stride = 5 * p.N * 8;
        if ((size_t)epb * p.stat_stride > stage_bytes) stage_bytes = (size_t)epb * p.stat_stride;
    }
    h->lds_bytes = (size_t)p.lds_stage + stage_bytes;
#ifdef FMARL_MEASURE
    if (const char *pad = getenv("FMARL_LDS_PAD")) h->lds_bytes += atoi(pad);   // lower occupancy
#endif
    h->grid = (p.n_envs + epb - 1) / epb;
    // navigation_graph's launches are store streams: scattered env blocks (cfg 3 one launch per step 1.525 -> 1.475 ms, 10 agents
    // 0.243 -> 0.216 ms on one box, spans into time slots unchanged: profiles/r4_scatter_ab.md).  The two formation scenarios'
    // launches are bound by their dependent chains and measured 1-2 % slower scattered: dispatch order.
    p.order = (form || fnav) ? 1 : scatter_order(h->grid);
    const uint64_t NEF = (uint64_t)p.N * p.E * p.F, maxq = (uint64_t)epb * NEF;
    if (maxq >= (1ull << 24) || maxq * NEF >= (1ull << 40)) { delete h; return fail(FMARL_EINVAL, "fmarl_create: shape too large"); }
    p.dNEF.set((uint32_t)NEF); p.dEF.set(p.E * p.F); p.dF.set(p.F); p.dEE.set(p.E * p.E); p.dE.set(p.E); p.dNE.set(p.N * p.E);
#ifdef FMARL_MEASURE
    if (const char *ab = getenv("FMARL_ABLATE")) p.ablate = atoi(ab);
#endif
    p.vec_node = !form && !fnav && (p.E * p.F) % 4 == 0 && p.E * p.F / 4 <= 64 * 4 && p.lds_env_bytes < 65536;
    p.vec_adj = p.E % 4 == 0 && p.has_posf;
    p.dC4.set(p.vec_node ? p.E * p.F / 4 : 1);
    p.dNC4.set(p.vec_node ? p.N * (p.E * p.F / 4) : 1);
    p.dEE4.set(p.vec_adj ? p.E * (p.E / 4) : 1);
    p.dE4.set(p.vec_adj ? p.E / 4 : 1);
    if (fnav) p.dC4.set(p.N * p.E);   // fairnav emission: (ego, entity) rows per env
    // small batches of navigation_graph (all agents of a workgroup in its first wave, rows of generic shape): waves 1 .. 3 emit while
    // wave 0 finishes the agents' part (step_body SMALL).  The statistics blocks, which share the emission windows' region, must sit
    // inside wave 0's window: the emission waves write the other three.
    const int partners = p.N + p.O + p.W;   // (every wave's window also takes its 64 contact pairs: 16 + 4 bytes each)
    h->small_ok = !form && !fnav && !p.vec_node && staged && epb * p.N <= 64 && partners <= 32 && epb * p.N * partners <= kThreads &&
                  p.stage_wave_bytes >= 64 * 20 && (!stat_shared || (size_t)epb * p.stat_stride <= (size_t)p.stage_wave_bytes);
    p.dP.set(partners); p.dN.set(p.N);
    {   // FastDiv (fmarl_dev.h) is exact only while dividend * divisor < 2^40: check every divisor against the largest
        // dividend its call sites form (indices inside one workgroup's share of an output array)
        const uint64_t E = p.E, N = p.N, F = p.F, EE = E * E, e = epb, lim = 1ull << 40;
        const bool ok = (e * EE + 16) * EE < lim && EE * E < lim && (e * N * E) * (N * E) < lim && (E * F) * F < lim &&
                        (e * N * E * F) * (E * F) < lim && (!p.vec_adj || (e * E * (E / 4)) * (E * (E / 4)) < lim) &&
                        (!p.vec_node || (e * N * (E * F / 4)) * (N * (E * F / 4)) < lim);
        if (!ok) { delete h; return fail(FMARL_EINVAL, "fmarl_create: shape too large (entity count beyond the index arithmetic of the emission)"); }
    }
    if (h->lds_bytes > 64 * 1024) {
        hipError_t e1 = hipSuccess;
        for (const void *f : {(const void *)step_kernel<0>, (const void *)step_kernel<2>, (const void *)step_end_kernel<0>, (const void *)step_end_kernel<1>,
                              (const void *)step_end_kernel<2>, (const void *)step_span_kernel<0>, (const void *)step_span_kernel<2>,
                              (const void *)step_small_kernel<0>, (const void *)step_small_kernel<1>,
                              (const void *)step_span_small_kernel<0>, (const void *)step_span_small_kernel<1>})
            if (e1 == hipSuccess) e1 = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
        for (const void *f : {(const void *)formation_span_kernel<0>, (const void *)formation_span_kernel<1>, (const void *)formation_kernel<true, 1>})
            if (e1 == hipSuccess) e1 = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
        hipError_t e2 = hipFuncSetAttribute((const void *)reset_emit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
        if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void *)rebuild_graph_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
        if (e1 == hipSuccess) e1 = hipFuncSetAttribute((const void *)formation_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
        if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void *)formation_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
        if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void *)formation_rebuild_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
        for (const void *f : {(const void *)fairnav_rebuild_kernel<256>, (const void *)fairnav_kernel<true, 256, 0>, (const void *)fairnav_kernel<true, 256, 3>, (const void *)fairnav_kernel<true, 256, 10>,
                              (const void *)fairnav_kernel<false, 256, 0>, (const void *)fairnav_span_kernel<192, 0>, (const void *)fairnav_span_kernel<192, 3>,
                              (const void *)fairnav_span_kernel<256, 0>, (const void *)fairnav_span_kernel<256, 3>})
            if (e2 == hipSuccess) e2 = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
        if (e1 != hipSuccess || e2 != hipSuccess) { delete h; return fail(FMARL_EHIP, "fmarl_create: cannot raise dynamic LDS limit"); }
    }
    h->place_lds = (size_t)(p.O + p.N + p.L) * 64 * sizeof(float2);
    if (h->place_lds > 150 * 1024) h->place_lds = 0;
    if (h->place_lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)reset_place_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)h->place_lds) != hipSuccess)
        h->place_lds = 0;
    h->async = (cfg->flags & FMARL_FLAG_ASYNC_RESET) && !form && !fnav;
    h->stage_dirty = true;   // nothing staged yet
    h->stage_pending = false;
    h->side = nullptr; h->ev_commit = h->ev_staged = nullptr;
    h->cap_id = h->cap_stage_id = 0;
    if (h->async) {
        int prio_lo = 0, prio_hi = 0;   // lowest priority: staging only fills what the step kernels leave idle
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if (hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio_lo) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_commit, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_staged, hipEventDisableTiming) != hipSuccess) {
            delete h;
            return fail(FMARL_EHIP, "fmarl_create: cannot create the staging stream / events");
        }
    }
    h->d_rot = nullptr;
    if (form && h->device >= 0) {   // the slots' rotation table: handle-owned device memory, filled once
        const int N = p.N;
        for (int i = 0; i < N; ++i) { const double a = i * ((2 * M_PI) / N); h->rot[2 * i] = cos(a); h->rot[2 * i + 1] = sin(a); }
        if (hipMalloc((void **)&h->d_rot, sizeof(double) * 2 * N) != hipSuccess ||
            hipMemcpy(h->d_rot, h->rot, sizeof(double) * 2 * N, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            if (h->d_rot) (void)hipFree(h->d_rot);
            h->d_rot = nullptr;   // (no usable device here: the handle still describes the layout; fmarl_init_state fills the state's copy)
        }
    }
    h->lockstep = false; h->host_step = 0; h->episode_started = false; h->captured = false;
    h->ev = nullptr; h->ev_steps = nullptr; h->ev_cap = h->ev_n = 0;
    memset(h->counts, 0, sizeof h->counts);
    *handle = h;
    return FMARL_OK;
}

static void drop_events(Handle *h) {
    for (int i = 0; h->ev && i < 2 * h->ev_cap; ++i) (void)hipEventDestroy(h->ev[i]);
    delete[] h->ev;
    delete[] h->ev_steps;
    h->ev = nullptr; h->ev_steps = nullptr; h->ev_cap = h->ev_n = 0;
}

int fmarl_destroy(void *handle) {
    Handle *h = (Handle *)handle;
    DeviceGuard on_device(h);
    if (h) {
        // (a handle may be destroyed -- e.g. by a garbage collector -- while some OTHER stream of the process is capturing:
        // the synchronisation below must not invalidate that capture)
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        drop_events(h);
        if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
        if (h->ev_commit) (void)hipEventDestroy(h->ev_commit);
        if (h->ev_staged) (void)hipEventDestroy(h->ev_staged);
        if (h->d_rot) (void)hipFree(h->d_rot);
        (void)hipThreadExchangeStreamCaptureMode(&mode);
    }
    delete h;
    return FMARL_OK;
}

int fmarl_envs_per_workgroup(void *handle) {
    Handle *h = (Handle *)handle;
    return h ? h->base.epb : 0;
}

int fmarl_launch_geometry(void *handle, int64_t *geometry) {
    Handle *h = (Handle *)handle;
    if (!h || !geometry) return fail(FMARL_EINVAL, "fmarl_launch_geometry: null argument");
    geometry[0] = h->grid; geometry[1] = h->threads; geometry[2] = (int64_t)h->lds_bytes; geometry[3] = h->base.epb;
    return FMARL_OK;
}

int fmarl_launch_counts(void *handle, int64_t *counts) {
    Handle *h = (Handle *)handle;
    if (!h || !counts) return fail(FMARL_EINVAL, "fmarl_launch_counts: null argument");
    memcpy(counts, h->counts, sizeof h->counts);
    return FMARL_OK;
}

int fmarl_profile_enable(void *handle, int capacity) {
    Handle *h = (Handle *)handle;
    DeviceGuard on_device(h);
    if (!h || capacity < 0) return fail(FMARL_EINVAL, "fmarl_profile_enable: bad argument");
    drop_events(h);
    if (capacity == 0) return FMARL_OK;
    h->ev = new (std::nothrow) hipEvent_t[2 * (size_t)capacity];
    h->ev_steps = new (std::nothrow) int[(size_t)capacity];
    if (!h->ev || !h->ev_steps) return fail(FMARL_EINVAL, "fmarl_profile_enable: out of host memory");
    for (int i = 0; i < 2 * capacity; ++i) HIP_OK(hipEventCreate(&h->ev[i]));
    h->ev_cap = capacity;
    return FMARL_OK;
}

int fmarl_profile_read(void *handle, float *ms, int *steps, int max_count, int *count) {
    Handle *h = (Handle *)handle;
    DeviceGuard on_device(h);
    if (!h || !ms || !count) return fail(FMARL_EINVAL, "fmarl_profile_read: bad argument");
    int n = h->ev_n < max_count ? h->ev_n : max_count;
    for (int i = 0; i < n; ++i) {
        HIP_OK(hipEventElapsedTime(&ms[i], h->ev[2 * i], h->ev[2 * i + 1]));
        if (steps) steps[i] = h->ev_steps[i];
    }
    *count = n;
    h->ev_n = 0;
    return FMARL_OK;
}

int fmarl_init_state(void *handle, void *state, void *stream) {
    Handle *h = (Handle *)handle;
    DeviceGuard on_device(h);
    if (!h || !state) return fail(FMARL_EINVAL, "fmarl_init_state: null argument");
    hipStream_t st = (hipStream_t)stream;
    h->cap_id = 0;
    HIP_OK(hipMemsetAsync(state, 0, h->layout.total, st));
    if (h->layout.count[FMARL_F_ROT_TABLE]) {
        const int N = h->cfg.num_agents;
        for (int i = 0; i < N; ++i) { const double a = i * ((2 * M_PI) / N); h->rot[2 * i] = cos(a); h->rot[2 * i + 1] = sin(a); }
        HIP_OK(hipMemcpyAsync((char *)state + h->layout.off[FMARL_F_ROT_TABLE], h->rot, sizeof(double) * 2 * N, hipMemcpyHostToDevice, st));
    }
    int rc = launch_reset(h, state, kResetInit, nullptr, nullptr, st);
    if (rc == FMARL_OK && h->async && h->cfg.scenario == FMARL_SCENARIO_NAVIGATION_GRAPH) {
        // The first launch of a kernel pays for loading its code (about a millisecond for the 50 KB of step_end_kernel), and the
        // first episode end may sit inside somebody's timed region: launch it once here over ZERO envs (one workgroup whose
        // threads are all inactive: no loads, no stores, the barriers only).
        Params q = bind(h, state);
        q.n_envs = 0;
        FmarlOutputs none = {};
        FMARL_NAV_END(h, step_end_kernel<SH>, dim3(1), dim3(kThreads), h->lds_bytes, st, q, none, (const int32_t *)nullptr,
                       (const float *)nullptr, 0);
        HIP_OK(hipGetLastError());
    }
    h->lockstep = !h->captured && h->cfg.scenario != FMARL_SCENARIO_FAIRNAV;   // fairnav episodes end early, env by env
    h->host_step = 0;
    return rc;
}

int fmarl_reset(void *handle, void *state, const uint8_t *env_mask, const FmarlOutputs *outs, void *stream) {
    Handle *h = (Handle *)handle;
    DeviceGuard on_device(h);
    if (!h || !state) return fail(FMARL_EINVAL, "fmarl_reset: null argument");
    if (outs && !outputs_aligned(h->base, outs)) return fail(FMARL_EINVAL, "fmarl_reset: node_obs / adj must be 16-byte aligned for this shape");
    h->cap_id = 0;
    if (h->async) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(FMARL_EINVAL, "fmarl_reset: not capturable on a handle with FMARL_FLAG_ASYNC_RESET (reset first, capture the steps)");
    }
    int rc = launch_reset(h, state, env_mask ? kResetMask : kResetAll, env_mask, outs, (hipStream_t)stream);
    if (env_mask || h->captured || h->cfg.scenario == FMARL_SCENARIO_FAIRNAV) h->lockstep = false; else { h->lockstep = true; h->host_step = 0; }
    h->episode_started = true;
    return rc;
}

int fmarl_step(void *handle, void *state, const int32_t *action_idx, const float *action_vec,
               const FmarlOutputs *outs, int auto_reset, void *stream) {
    Handle *h = (Handle *)handle;
    DeviceGuard on_device(h);
    if (!h || !state || !outs) return fail(FMARL_EINVAL, "fmarl_step: null argument");
    if ((action_idx == nullptr) == (action_vec == nullptr))
        return fail(FMARL_EINVAL, "fmarl_step: pass exactly one of action_idx / action_vec");
    hipStream_t st = (hipStream_t)stream;
    Params p = bind(h, state);
    if (!outputs_aligned(p, outs)) return fail(FMARL_EINVAL, "fmarl_step: node_obs / adj must be 16-byte aligned for this shape");
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    unsigned long long cap_id = 0;
    h->cap_id = 0;
    if (hipStreamGetCaptureInfo(st, &cs, &cap_id) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        if (auto_reset == FMARL_RESET_LOCKSTEP) {
            // the caller vouches for the replay phase (include/fmarl.h): decide from the host mirror like an eager step.
            // With the staged reset the side stream becomes a forked branch of the graph (launch_stage).
            if (!h->lockstep) return fail(FMARL_EINVAL, "fmarl_step: FMARL_RESET_LOCKSTEP capture needs envs in lockstep (fmarl_get_phase() >= 0)");
            h->cap_id = cap_id ? cap_id : 1;
            if (h->async && !h->stage_pending && h->cap_stage_id != h->cap_id)
                return fail(FMARL_EINVAL, "fmarl_step: capture a handle with FMARL_FLAG_ASYNC_RESET from the first step after a reset "
                                          "(the staging of the next episode must be part of the graph)");
        } else {
            // Nothing runs during capture and a replay can start at any phase of an episode, any number of times: the
            // reset-or-not decision must not be baked from the host's mirror of the step counter.  Every captured step
            // enqueues the auto-reset launches, which test cur_step per env on the device.  Those are the synchronous
            // reset's launches: a handle with the staged reset becomes a synchronous one from here on (same results).
            h->captured = true;
            h->lockstep = false;
            if (h->async) { h->async = false; h->stage_pending = false; h->stage_dirty = true; }
        }
    }
    if (h->stage_pending) {   // first step after a reset: stage the episode after this one on the side stream
        h->stage_pending = false;
        int rc = launch_stage(h, state, st);
        if (rc) return rc;
    }
    // navigation_graph, envs in lockstep, the step that ends the episode, the next episode staged: one launch does the step,
    // the commit and the reset observation (step_end_kernel) instead of step + reset_commit + reset_emit
    const bool fold = auto_reset && p.scenario == FMARL_SCENARIO_NAVIGATION_GRAPH && h->async && (h->cap_id ? h->cap_stage_id == h->cap_id : !h->stage_dirty) && h->lockstep &&
                      h->host_step + 1 >= h->cfg.episode_length && (outs->obs || outs->node_obs || outs->adj);
    // (a captured step reaches this point only with the staging of this capture in place: the test above)
    if (fold) HIP_OK(hipStreamWaitEvent(st, h->ev_staged, 0));   // the staged episode must be complete
    const bool prof = h->ev && h->ev_n < h->ev_cap && !h->cap_id && cs == hipStreamCaptureStatusNone;
    if (prof) HIP_OK(hipEventRecord(h->ev[2 * h->ev_n], st));
    if (p.scenario == FMARL_SCENARIO_FAIRNAV)
        FMARL_FNAV_NL(h, (fairnav_kernel<true, TH, NL>), dim3(h->grid), h->lds_bytes, st, p, *outs, action_idx, action_vec, auto_reset ? 1 : 0);
    else if (p.scenario == FMARL_SCENARIO_FORMATION)
        FMARL_FORM(h, (formation_kernel<true, SH>), dim3(h->grid), dim3(h->threads), h->lds_bytes, st, p, *outs, action_idx,
                   action_vec, auto_reset ? 1 : 0);
    else if (fold)
        FMARL_NAV_END(h, step_end_kernel<SH>, dim3(h->grid), dim3(kThreads), h->lds_bytes, st, p, *outs, action_idx, action_vec, 1);
    else if (h->small_ok && !outs->edge_nnz)
        FMARL_NAV_SMALL(h, step_small_kernel<SH>, dim3(h->grid), dim3(kThreads), h->lds_bytes, st, p, *outs, action_idx, action_vec,
                        auto_reset ? 1 : 0);
    else
        FMARL_NAV_FULL(h, step_kernel<SH>, dim3(h->grid), dim3(kThreads), h->lds_bytes, st, p, *outs, action_idx, action_vec,
                           auto_reset ? 1 : 0);
    if (prof) { HIP_OK(hipEventRecord(h->ev[2 * h->ev_n + 1], st)); h->ev_steps[h->ev_n] = 1; ++h->ev_n; }
    HIP_OK(hipGetLastError());
    ++h->counts[0];
    if (fold) ++h->counts[1];
    if (h->lockstep) ++h->host_step;
    h->episode_started = false;
    if (auto_reset && p.scenario == FMARL_SCENARIO_FAIRNAV) {
        h->episode_started = true;   // this scenario's episodes end env by env: fairnav_kernel<true> resets them itself
    } else if (auto_reset) {
        const bool may_reset = !h->lockstep || h->host_step >= h->cfg.episode_length;
        if (fold) {   // committed and emitted by step_end_kernel; the episode after this one is staged by the next step
            h->host_step = 0;
            h->episode_started = true;
            h->stage_pending = true;
        } else if (may_reset) {
            int rc = launch_reset(h, state, kResetAuto, nullptr, outs, st);
            if (rc) return rc;
            ++h->counts[2];
            if (h->lockstep) h->host_step = 0;
            h->episode_started = true;
        }
    }
    return FMARL_OK;
}

int fmarl_step_span(void *handle, void *state, const int32_t *action_idx, const float *action_vec, int n_steps,
                    const FmarlOutputs *outs, const FmarlSpan *span, void *stream) {
    Handle *h = (Handle *)handle;
    if (!h || !state || !outs || !span || n_steps < 0) return fail(FMARL_EINVAL, "fmarl_step_span: bad argument");
    if ((action_idx == nullptr) == (action_vec == nullptr))
        return fail(FMARL_EINVAL, "fmarl_step_span: pass exactly one of action_idx / action_vec");
    const int sc = h->cfg.scenario;
    hipStream_t st = (hipStream_t)stream;
    {   // a span decides on the host where episodes end: not for stream capture (capture fmarl_step calls instead)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(FMARL_EINVAL, "fmarl_step_span: not capturable (capture fmarl_step calls with FMARL_RESET_LOCKSTEP instead)");
    }
    SpanStrides s = {span->obs, span->node_obs, span->adj, span->reward, span->done, span->info, span->edge_nnz, span->graph_record, span->actions};
    int t = 0;
    while (t < n_steps) {
        // steps that certainly end no episode go out as one span launch; the step that ends an episode and envs out of
        // lockstep go through fmarl_step.  The third scenario's episodes end env by env and its step resets them itself: all
        // its steps are one launch (fairnav_span_kernel: the state through global memory between the steps).
        int k = 0;
        // (its span carries the step counter in 15 bits and the two collision counts in 16 each -- FairnavCarry: episodes too long for
        // that go out one launch per step, the state through global memory, same results)
        // Nor do more than three agents: the step is then bound by the assignment and the sequential walk, whose waves want every register,
        // and only a launch per step can start its assignment from the previous step's (the span's emission windows run over the match
        // table) -- measured per step at 65 536 envs, one launch per step / the span: N = 3 0.0483 / 0.0403 ms, 4 0.0961 / 0.1010,
        // 5 0.1437 / 0.159, 6 0.2143 / 0.2357, 8 0.3773 / 0.3838, 10 0.634 / 0.668 (profiles/r6_fnav_spans_by_n.txt).
        if (sc == FMARL_SCENARIO_FAIRNAV)
            k = (h->cfg.num_agents <= 3 && h->cfg.episode_length < 32768 && (long long)(h->cfg.num_agents - 1) * h->cfg.episode_length <= 65535) ? n_steps - t : 0;
        else if (h->lockstep) k = h->cfg.episode_length - 1 - h->host_step;
        if (k > n_steps - t) k = n_steps - t;
        FmarlOutputs o = *outs;
        if (o.obs) o.obs += (size_t)t * span->obs;
        if (o.node_obs) o.node_obs += (size_t)t * span->node_obs;
        if (o.adj) o.adj += (size_t)t * span->adj;
        if (o.reward) o.reward += (size_t)t * span->reward;
        if (o.done) o.done += (size_t)t * span->done;
        if (o.info) o.info += (size_t)t * span->info;
        if (o.edge_nnz) o.edge_nnz += (size_t)t * span->edge_nnz;
        if (o.graph_record) o.graph_record += (size_t)t * span->graph_record;
        const int32_t *a = action_idx ? action_idx + (size_t)t * span->actions : nullptr;
        const float *av = action_vec ? action_vec + (size_t)t * span->actions : nullptr;
        if (k >= 2) {
            DeviceGuard on_device(h);
            if (h->stage_pending) {   // the span starts an episode: stage the one after it on the side stream first (as fmarl_step would)
                h->stage_pending = false;
                h->cap_id = 0;
                int rc = launch_stage(h, state, st);
                if (rc) return rc;
            }
            Params p = bind(h, state);
            if (!outputs_aligned(p, &o)) return fail(FMARL_EINVAL, "fmarl_step_span: node_obs / adj must be 16-byte aligned for this shape");
            const bool prof = h->ev && h->ev_n < h->ev_cap;
            if (prof) HIP_OK(hipEventRecord(h->ev[2 * h->ev_n], st));
            if (sc == FMARL_SCENARIO_FAIRNAV)
                FMARL_FNAV_T(h, h->span_threads, (fairnav_span_kernel<TH, NL>), dim3(h->grid), h->lds_bytes, st, FairnavSpanArgs{p, o, s, a, av, k, 1});
            else if (sc == FMARL_SCENARIO_FORMATION)
                FMARL_FORM(h, formation_span_kernel<SH>, dim3(h->grid), dim3(h->threads), h->lds_bytes, st, p, o, s, a, av, k);
            else if (h->small_ok && !o.edge_nnz)
                FMARL_NAV_SMALL(h, step_span_small_kernel<SH>, dim3(h->grid), dim3(h->threads), h->lds_bytes, st, p, o, s, a, av, k);
            else
                FMARL_NAV_FULL(h, step_span_kernel<SH>, dim3(h->grid), dim3(h->threads), h->lds_bytes, st, p, o, s, a, av, k);
            if (prof) { HIP_OK(hipEventRecord(h->ev[2 * h->ev_n + 1], st)); h->ev_steps[h->ev_n] = k; ++h->ev_n; }
            HIP_OK(hipGetLastError());
            if (h->lockstep) h->host_step += k;
            h->episode_started = sc == FMARL_SCENARIO_FAIRNAV;   // (as fmarl_step: that scenario's envs start their episodes inside the step)
            h->counts[0] += k;
            t += k;
        } else {
            int rc = fmarl_step(handle, state, a, av, &o, FMARL_RESET_AUTO, stream);
            if (rc) return rc;
            ++t;
        }
    }
    return FMARL_OK;
}

static int copy_field(Handle *h, void *state, int field, void *host_or_dev, bool to_state, hipStream_t st, const char *who) {
    if (!h || !state || !host_or_dev) return fail(FMARL_EINVAL, "%s: null argument", who);
    DeviceGuard on_device(h);
    if (field < 0 || field >= FMARL_NUM_FIELDS) return fail(FMARL_EINVAL, "%s: bad field id", who);
    const size_t bytes = h->layout.count[field] * dtype_bytes(h->layout.dtype[field]);
    if (bytes == 0) return FMARL_OK;
    char *f = (char *)state + h->layout.off[field];
    HIP_OK(hipMemcpyAsync(to_state ? (void *)f : host_or_dev, to_state ? host_or_dev : (void *)f, bytes, hipMemcpyDefault, st));
    return FMARL_OK;
}

int fmarl_get_state(void *handle, const void *state, int field, void *dst, void *stream) {
    return copy_field((Handle *)handle, (void *)state, field, dst, false, (hipStream_t)stream, "fmarl_get_state");
}

int fmarl_set_state(void *handle, void *state, int field, const void *src, void *stream) {
    int rc = copy_field((Handle *)handle, state, field, (void *)src, true, (hipStream_t)stream, "fmarl_set_state");
    if (rc == FMARL_OK) rc = fmarl_state_changed(handle);
    return rc;
}

int fmarl_poison_lds(void *handle, void *stream) {
    Handle *h = (Handle *)handle;
    if (!h) return fail(FMARL_EINVAL, "fmarl_poison_lds: null handle");
    DeviceGuard on_device(h);
    const int bytes = 64 * 1024;   // two such workgroups fit the 160 KB of a CU; a few rounds of them reach every CU's whole LDS
    hipLaunchKernelGGL(fmarl::poison_lds_kernel, dim3(256 * 8 * 4), dim3(256), bytes, (hipStream_t)stream, bytes / 4);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_store_stream(void *dst, size_t bytes, int shape, size_t chunk_bytes, int order, int persist, void *stream) {
    if (!dst || ((uintptr_t)dst & 15) || (bytes & 15) || shape < 0 || shape > 4 || order < 1 || persist < 0)
        return fail(FMARL_EINVAL, "fmarl_store_stream: dst / bytes must be 16-byte multiples, shape 0..4, order >= 1, persist >= 0");
    if (shape != 0 && (chunk_bytes < 4096 || (chunk_bytes & 15) || chunk_bytes > ((size_t)1 << 34)))
        return fail(FMARL_EINVAL, "fmarl_store_stream: chunk_bytes must be a 16-byte multiple of at least 4096");
    const size_t n16 = bytes / 16, c16 = shape ? chunk_bytes / 16 : 0;
    if (!n16) return FMARL_OK;
    const size_t chunks = shape ? (n16 + c16 - 1) / c16 : 0;
    if (chunks > 0x7fffffffu) return fail(FMARL_EINVAL, "fmarl_store_stream: too many chunks");
    if (shape) {   // the chunk order must be a permutation: order coprime with the number of chunks
        size_t a = chunks, b = (size_t)order % chunks;
        while (b) { const size_t r = a % b; a = b; b = r; }
        if (chunks > 1 && a != 1) return fail(FMARL_EINVAL, "fmarl_store_stream: order must be coprime with the number of chunks");
    }
    const size_t grid = shape ? (persist && (size_t)persist < chunks ? (size_t)persist : chunks) : 2048;
    hipStream_t st = (hipStream_t)stream;
    if (shape == 0) hipLaunchKernelGGL(fmarl::store_stream_kernel<0>, dim3((unsigned)grid), dim3(256), 0, st, (float4 *)dst, n16, 0u, 0u, 1u);
    else if (shape == 1) hipLaunchKernelGGL(fmarl::store_stream_kernel<1>, dim3((unsigned)grid), dim3(256), 0, st, (float4 *)dst, n16, (uint32_t)c16, (uint32_t)chunks, (uint32_t)order);
    else if (shape == 2) hipLaunchKernelGGL(fmarl::store_stream_kernel<2>, dim3((unsigned)grid), dim3(256), 0, st, (float4 *)dst, n16, (uint32_t)c16, (uint32_t)chunks, (uint32_t)order);
    else if (shape == 3) hipLaunchKernelGGL(fmarl::store_stream_kernel<3>, dim3((unsigned)grid), dim3(256), 0, st, (float4 *)dst, n16, (uint32_t)c16, (uint32_t)chunks, (uint32_t)order);
    else hipLaunchKernelGGL(fmarl::store_stream_kernel<4>, dim3((unsigned)grid), dim3(256), 0, st, (float4 *)dst, n16, (uint32_t)c16, (uint32_t)chunks, (uint32_t)order);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_store_pattern(void *node, void *adj, size_t node_group_bytes, size_t adj_group_bytes, int groups, int slots, size_t node_slot_bytes,
                        size_t adj_slot_bytes, int window_bytes, int order, void *stream) {
    if (!node || !adj || groups < 1 || slots < 1 || window_bytes < 64 || (window_bytes & 3) || (node_group_bytes & 3) || (adj_group_bytes & 3) ||
        (node_slot_bytes & 3) || (adj_slot_bytes & 3) || order < 1 || (((uintptr_t)node | (uintptr_t)adj) & 3))
        return fail(FMARL_EINVAL, "fmarl_store_pattern: bad argument (sizes are multiples of 4 bytes, window_bytes >= 64)");
    if ((size_t)groups * node_group_bytes < node_slot_bytes || (size_t)groups * adj_group_bytes < adj_slot_bytes)
        return fail(FMARL_EINVAL, "fmarl_store_pattern: the groups do not cover a slot");
    {
        size_t a = (size_t)groups, b = (size_t)order % (size_t)groups;
        while (b) { const size_t r = a % b; a = b; b = r; }
        if (groups > 1 && a != 1) return fail(FMARL_EINVAL, "fmarl_store_pattern: order must be coprime with the number of groups");
    }
    hipLaunchKernelGGL(fmarl::store_pattern_kernel, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream, (char *)node, (char *)adj, node_group_bytes,
                       adj_group_bytes, (uint32_t)groups, (uint32_t)slots, node_slot_bytes, adj_slot_bytes, (uint32_t)window_bytes, (uint32_t)order);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

// ---- time-slot arrays with interleaved physical memory (include/fmarl.h fmarl_ring_alloc)
namespace {
// A virtual address range this process reserved for an array of pieces.  An address that has carried a mapping is NEVER used again,
// neither by the runtime (hipMemAddressFree) nor by this allocator: on this stack (ROCm 7.2, MI355X) the GPU keeps translations of
// an unmapped range for a while.  Round 4 found it with ranges that went back to the runtime and came out of a later
// hipMemAddressReserve again (an array mapped there read back zeroes in up to 70 % of its bytes right after a fill, other bytes
// changed seconds later: tools/vmm_reuse_probe.py); round 5 tried to re-use a KEPT range for the next array of the same size --
// fresh pieces mapped into a range that never left the process -- and hit the same fault: the new array passed a fill / read-back
// through the copy engines and then lost 20 MiB of a kernel's fill (tests/test_hip_parity.py
// test_time_slots_after_a_freed_array_keep_what_is_written, third array; profiles/r5_notes.md).  Writes through a stale translation land in
// physical memory that has been released: not a path to keep "verified".  So freed ranges stay reserved and idle.  What that costs is
// address space, and it is bounded and visible: the reservations are counted (fmarl_ring_stats) and a request that would take them
// past the cap -- 8 TiB of the 47 bits, FMARL_RING_RESERVE_CAP_GB overrides -- is refused with an error: allocate plainly then.
struct RingAlloc {
    void *ptr = nullptr;
    size_t total = 0, piece = 0, mapped = 0;   // `mapped` virtual pieces from the start of the range are mapped
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
std::mutex g_ring_mutex;
uint64_t g_ring_reserved = 0, g_ring_live = 0, g_ring_ranges = 0, g_ring_live_ranges = 0, g_ring_refused = 0;
uint64_t g_ring_frees = 0, g_ring_checked = 0, g_ring_check_failed = 0;   // arrays freed; arrays checked by a kernel's fill; of them failed
uint64_t ring_reserve_cap() {
    static const uint64_t cap = [] {
        const char *e = getenv("FMARL_RING_RESERVE_CAP_GB");
        const double gb = e ? atof(e) : 0.0;
        return gb > 0.0 ? (uint64_t)(gb * 1073741824.0) : (uint64_t)8 << 40;
    }();
    return cap;
}

void ring_release(RingAlloc *r, bool was_handed_out = true) {
    if (!r) return;
    // no launch may still write here: the library waits for the device itself rather than trust every caller to have done so (a
    // destructor that runs during somebody's stream capture cannot synchronise: the error is dropped, the pieces go back anyway)
    if (r->mapped && hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
    if (r->ptr && r->mapped == 0) {
        // a range that never carried a mapping (a piece could not be created: out of memory) has no translations anywhere: it goes back
        // to the runtime and off the books -- failed attempts (retries with fewer slots) must not eat the reserve cap
        (void)hipMemAddressFree(r->ptr, r->total);
        for (auto h : r->handles) (void)hipMemRelease(h);
        std::lock_guard<std::mutex> lock(g_ring_mutex);
        g_ring_reserved -= r->total; g_ring_live -= r->total; --g_ring_ranges; --g_ring_live_ranges;
        delete r;
        return;
    }
    for (size_t k = 0; r->ptr && k < r->mapped; ++k) (void)hipMemUnmap((char *)r->ptr + k * r->piece, r->piece);   // (piece by piece, as they were mapped)
    for (auto h : r->handles) (void)hipMemRelease(h);
    if (r->ptr) {   // the range stays reserved, idle from here on
        std::lock_guard<std::mutex> lock(g_ring_mutex);
        g_ring_live -= r->total; --g_ring_live_ranges;
        if (was_handed_out) ++g_ring_frees;
    }
    delete r;
}

// An array allocated after another one of this process was freed -- the only condition either fault was ever seen under (re-reserved
// ranges: round 4; kept ranges with fresh pieces: round 5; tools/vmm_fault_repro.cpp shows both on the bare HIP calls) -- is checked
// before it is handed out: a kernel fills it with a pattern of the word index, a second kernel reads it back, counts the words that
// do not hold it and leaves zeroes.  Two passes over the array (205 GB: 60 ms), never for a process's first arrays.
// -> number of wrong 32-bit words, or -1 when the check itself could not run.
long long ring_kernel_check(void *ptr, size_t total) {
    unsigned long long *bad = nullptr, host = 0;
    if (hipMalloc((void **)&bad, sizeof(*bad)) != hipSuccess) { (void)hipGetLastError(); return -1; }
    const size_t n16 = total / 16;
    const uint32_t salt = 0x9e3779b9u ^ (uint32_t)(uintptr_t)ptr ^ (uint32_t)((uintptr_t)ptr >> 32);
    hipError_t e = hipMemset(bad, 0, sizeof(*bad));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fmarl::ring_fill_kernel, dim3(4096), dim3(256), 0, (hipStream_t)0, (uint4 *)ptr, n16, salt);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fmarl::ring_check_kernel, dim3(4096), dim3(256), 0, (hipStream_t)0, (uint4 *)ptr, n16, salt, bad);
        e = hipMemcpy(&host, bad, sizeof(host), hipMemcpyDeviceToHost);
    }
    (void)hipFree(bad);
    if (e != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (long long)host;
}
}  // namespace

int fmarl_ring_alloc(size_t slot_bytes, int slots, size_t piece_bytes, void **base, void **cookie) {
    if (!base || !cookie || slots < 1 || slot_bytes == 0) return fail(FMARL_EINVAL, "fmarl_ring_alloc: bad argument");
    *base = *cookie = nullptr;
    int dev = 0, vmm = 0;
    HIP_OK(hipGetDevice(&dev));
    HIP_OK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    if (!vmm) return fail(FMARL_EHIP, "fmarl_ring_alloc: the device has no virtual memory management");
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0;
    HIP_OK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (gran < 4096) gran = 4096;
    if (piece_bytes == 0) {
        // the largest divisor of a slot that is a multiple of the granularity and at most 16 MiB (a slot = k pieces, k as small as
        // possible: every piece is a hipMemCreate call, and those get slow in the tens of thousands); 0 pieces found = no divisor
        const size_t cap = (size_t)16 << 20;
        for (size_t k = (slot_bytes + cap - 1) / cap; k <= slot_bytes / gran && k <= 65536; ++k)
            if (slot_bytes % k == 0 && (slot_bytes / k) % gran == 0) { piece_bytes = slot_bytes / k; break; }
        if (piece_bytes == 0) return fail(FMARL_EINVAL, "fmarl_ring_alloc: the slot size has no divisor that is a multiple of the allocation granularity");
    }
    if (piece_bytes < gran || piece_bytes % gran || slot_bytes % piece_bytes)
        return fail(FMARL_EINVAL, "fmarl_ring_alloc: a slot must be a whole number of pieces, a piece a multiple of the allocation granularity");
    const size_t per_slot = slot_bytes / piece_bytes, count = per_slot * (size_t)slots, total = slot_bytes * (size_t)slots;
    RingAlloc *r = new (std::nothrow) RingAlloc();
    if (!r) return fail(FMARL_EINVAL, "fmarl_ring_alloc: out of host memory");
    r->total = total;
    r->piece = piece_bytes;
    {
        std::lock_guard<std::mutex> lock(g_ring_mutex);
        if (g_ring_reserved + total > ring_reserve_cap()) {
            ++g_ring_refused;
            delete r;
            return fail(FMARL_EINVAL, "fmarl_ring_alloc: the address space this process may reserve for arrays of pieces is used up (freed arrays keep "
                                      "their ranges: the GPU holds on to translations of unmapped addresses; FMARL_RING_RESERVE_CAP_GB, default 8192): allocate plainly");
        }
        hipError_t e = hipMemAddressReserve(&r->ptr, total, (size_t)2 << 20, nullptr, 0);   // (the alignment must be a power of two; pieces need not be)
        if (e != hipSuccess) {
            (void)hipGetLastError();   // (not left behind for the caller's next HIP call to trip over)
            delete r;
            return fail(FMARL_EHIP, "fmarl_ring_alloc: hipMemAddressReserve: %s", hipGetErrorString(e));
        }
        g_ring_reserved += total; g_ring_live += total; ++g_ring_ranges; ++g_ring_live_ranges;
    }
    void *ptr = r->ptr;
    hipError_t e = hipSuccess;
    r->handles.reserve(count);
    for (size_t k = 0; k < count && e == hipSuccess; ++k) {   // physical pieces, in creation order
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, piece_bytes, &prop, 0);
        if (e == hipSuccess) r->handles.push_back(h);
    }
    // virtual piece j of slot t <- physical piece j * slots + t: a slot's pieces are spread evenly over the whole allocation
    for (size_t t = 0; t < (size_t)slots && e == hipSuccess; ++t)
        for (size_t j = 0; j < per_slot && e == hipSuccess; ++j) {
            e = hipMemMap((char *)ptr + (t * per_slot + j) * piece_bytes, piece_bytes, 0, r->handles[j * (size_t)slots + t], 0);
            if (e == hipSuccess) ++r->mapped;
        }
    if (e == hipSuccess) {
        // read / write for this device and for every device that has peer access to it (a learner rank's copy out of a time slot,
        // tensor.to('cuda:1')): hipMalloc memory is peer-accessible once peer access is enabled, an array of pieces only for the
        // devices named here.  Should the runtime refuse the peers, the array is still this device's.
        std::vector<hipMemAccessDesc> acc(1);
        acc[0].location.type = hipMemLocationTypeDevice; acc[0].location.id = dev; acc[0].flags = hipMemAccessFlagsProtReadWrite;
        // (opt-in, FMARL_RING_PEER_ACCESS=1: mapping a 200 GB ring into seven more devices' page tables is not something every
        // process of a job should pay for, and no multi-GPU box has exercised it yet; without it a peer's read of a time slot goes
        // through a staging copy on this device -- the trajectory exchange of bench.py never reads peers' slots)
        int ndev = 0;
        const char *peer_env = getenv("FMARL_RING_PEER_ACCESS");
        if (!(peer_env && peer_env[0] == '1')) ndev = 0;
        else if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); ndev = 0; }
        for (int d = 0; d < ndev; ++d) {
            int can = 0;
            if (d != dev && hipDeviceCanAccessPeer(&can, d, dev) == hipSuccess && can) {
                hipMemAccessDesc a = acc[0]; a.location.id = d; acc.push_back(a);
            }
        }
        e = hipMemSetAccess(ptr, total, acc.data(), acc.size());
        if (e != hipSuccess && acc.size() > 1) { (void)hipGetLastError(); e = hipMemSetAccess(ptr, total, acc.data(), 1); }
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();   // (not left behind for the caller's next HIP call to trip over)
        const hipError_t why = e;
        ring_release(r, false);    // unmaps what was mapped, releases every piece; a range that carried a mapping stays reserved (and counted)
        return fail(FMARL_EHIP, "fmarl_ring_alloc: %s", hipGetErrorString(why));
    }
    bool check;
    {
        std::lock_guard<std::mutex> lock(g_ring_mutex);
        check = g_ring_frees > 0;
    }
    if (const char *v = getenv("FMARL_RING_VERIFY")) check = v[0] == '1' ? true : (v[0] == '0' ? false : check);   // 1: always, 0: never
    if (check) {
        const long long bad = ring_kernel_check(ptr, total);
        {
            std::lock_guard<std::mutex> lock(g_ring_mutex);
            ++g_ring_checked;
            if (bad != 0) ++g_ring_check_failed;
        }
        if (bad != 0) {
            fprintf(stderr, "libfmarl: fmarl_ring_alloc: an array of %zu bytes allocated after an earlier one was freed did not hold a kernel's fill "
                            "(%lld wrong words); refused -- allocate plainly\n", total, bad);
            ring_release(r, false);
            char words[32];
            snprintf(words, sizeof(words), "%lld", bad);
            return fail(FMARL_EHIP, "fmarl_ring_alloc: the array did not hold a kernel's fill (%s wrong words): allocate plainly", words);
        }
    }
    *base = ptr; *cookie = r;
    return FMARL_OK;
}

int fmarl_ring_free(void *cookie) {
    if (!cookie) return fail(FMARL_EINVAL, "fmarl_ring_free: null cookie");
    ring_release((RingAlloc *)cookie);
    return FMARL_OK;
}

int fmarl_ring_stats(uint64_t out[8]) {
    if (!out) return fail(FMARL_EINVAL, "fmarl_ring_stats: null argument");
    std::lock_guard<std::mutex> lock(g_ring_mutex);
    out[0] = g_ring_reserved; out[1] = g_ring_reserved - g_ring_live; out[2] = g_ring_ranges; out[3] = g_ring_ranges - g_ring_live_ranges;
    out[4] = ring_reserve_cap(); out[5] = g_ring_refused; out[6] = g_ring_checked; out[7] = g_ring_check_failed;
    return FMARL_OK;
}

int fmarl_insert_masks(const uint8_t *done, float *masks, float *active_masks, int64_t rows, int num_agents, void *stream) {
    if (!done || !masks || !active_masks || rows < 0 || num_agents < 1) return fail(FMARL_EINVAL, "fmarl_insert_masks: bad argument");
    if (!rows) return FMARL_OK;
    hipLaunchKernelGGL(fmarl::insert_masks_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, done, masks,
                       active_masks, (size_t)rows, num_agents);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_get_phase(void *handle) {
    Handle *h = (Handle *)handle;
    return h && h->lockstep ? h->host_step : -1;
}

int fmarl_set_phase(void *handle, int phase) {
    Handle *h = (Handle *)handle;
    if (!h || !h->lockstep || phase < 0 || phase >= h->cfg.episode_length) return fail(FMARL_EINVAL, "fmarl_set_phase: envs not in lockstep or bad phase");
    h->host_step = phase;
    return FMARL_OK;
}

int fmarl_state_changed(void *handle) {
    Handle *h = (Handle *)handle;
    if (!h) return fail(FMARL_EINVAL, "fmarl_state_changed: null handle");
    h->lockstep = false;
    h->stage_dirty = true;   // seed / episode counters may have changed: restage before the next commit
    return FMARL_OK;
}

int fmarl_cost_matrix(const double *agent_pos, const double *goal_pos, double *costs, int n_envs, int num_agents,
                      int num_goals, void *stream) {
    if (!agent_pos || !goal_pos || !costs || n_envs < 1 || num_agents < 1 || num_goals < 1)
        return fail(FMARL_EINVAL, "fmarl_cost_matrix: bad argument");
    const size_t total = (size_t)n_envs * num_agents * num_goals;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(cost_matrix_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double2 *)agent_pos,
                       (const double2 *)goal_pos, costs, n_envs, num_agents, num_goals);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_lexifair(const double *costs, int32_t *perm, int n_envs, int num_agents, void *stream) {
    if (!costs || !perm || n_envs < 1 || num_agents < 1 || num_agents > 64)
        return fail(FMARL_EINVAL, "fmarl_lexifair: bad argument (1 <= N <= 64)");
    launch_lexifair_costs(costs, perm, n_envs, num_agents, (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_update_graph(const float *adj, int32_t *edge_index, float *edge_weight, int32_t *nnz, int n_envs,
                       int num_entities, double max_edge_dist, void *stream) {
    if (!adj || !edge_index || !edge_weight || !nnz || n_envs < 1 || num_entities < 1)
        return fail(FMARL_EINVAL, "fmarl_update_graph: bad argument");
    const int blocks = (n_envs + 3) / 4;   // 4 waves per workgroup, one env per wave
    hipLaunchKernelGGL(update_graph_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, adj, edge_index,
                       edge_weight, nnz, n_envs, num_entities, (float)max_edge_dist);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_update_graph_state(void *handle, const void *state, int32_t *edge_index, double *edge_weight, int32_t *nnz,
                             void *stream) {
    Handle *h = (Handle *)handle;
    DeviceGuard on_device(h);
    if (!h || !state || !edge_index || !edge_weight || !nnz) return fail(FMARL_EINVAL, "fmarl_update_graph_state: null argument");
    Params p = bind(h, (void *)state);
    hipLaunchKernelGGL(update_graph_state_kernel, dim3((p.n_envs + 3) / 4), dim3(256), 0, (hipStream_t)stream, p, edge_index,
                       edge_weight, nnz, h->cfg.max_edge_dist);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_edge_count(const float *adj, int32_t *nnz, int n_envs, int num_entities, double max_edge_dist, int strict,
                     void *stream) {
    if (!adj || !nnz || n_envs < 1 || num_entities < 1) return fail(FMARL_EINVAL, "fmarl_edge_count: bad argument");
    hipLaunchKernelGGL(edge_count_kernel, dim3((n_envs + 3) / 4), dim3(256), 0, (hipStream_t)stream, adj, nnz, n_envs,
                       num_entities, (float)max_edge_dist, strict);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_edge_fill(const float *adj, const int64_t *offsets, int64_t *edge_index, float *edge_attr, int64_t total,
                    int n_graphs, int graphs_per_env, int num_entities, double max_edge_dist, int strict, void *stream) {
    if (!adj || !offsets || !edge_index || !edge_attr || n_graphs < 1 || graphs_per_env < 1 || num_entities < 1 || total < 0)
        return fail(FMARL_EINVAL, "fmarl_edge_fill: bad argument");
    hipLaunchKernelGGL(edge_fill_kernel, dim3((n_graphs + 3) / 4), dim3(256), 0, (hipStream_t)stream, adj, offsets,
                       edge_index, edge_attr, total, n_graphs, graphs_per_env, num_entities, (float)max_edge_dist, strict);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_edge_offsets(const int32_t *nnz, int n_envs, int graphs_per_env, int64_t *offsets, void *stream) {
    if (!nnz || !offsets || n_envs < 1 || graphs_per_env < 1 || (int64_t)n_envs * graphs_per_env > 0x7fffffff)
        return fail(FMARL_EINVAL, "fmarl_edge_offsets: bad argument");
    const int n_graphs = n_envs * graphs_per_env, chunks = (n_graphs + kScanChunk - 1) / kScanChunk;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(edge_scan_totals_kernel, dim3(chunks), dim3(256), 0, st, nnz, n_graphs, graphs_per_env, offsets);
    hipLaunchKernelGGL(edge_scan_chunks_kernel, dim3(1), dim3(256), 0, st, n_graphs, offsets);
    hipLaunchKernelGGL(edge_scan_fill_kernel, dim3(chunks), dim3(256), 0, st, nnz, n_graphs, graphs_per_env, offsets);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_edge_fill_state(void *handle, const void *state, const int64_t *offsets, int64_t *edge_index, float *edge_attr,
                          int64_t capacity, int graphs_per_env, int32_t *mismatch, void *stream) {
    Handle *h = (Handle *)handle;
    DeviceGuard on_device(h);
    if (!h || !state || !offsets || capacity < 0 || graphs_per_env < 1 || (capacity > 0 && (!edge_index || !edge_attr)))
        return fail(FMARL_EINVAL, "fmarl_edge_fill_state: bad argument");
    if (capacity == 0) return FMARL_OK;
    Params p = bind(h, (void *)state);
    const int64_t graphs = (int64_t)p.n_envs * graphs_per_env;
    if (graphs > 0x7fffffff / 64) return fail(FMARL_EINVAL, "fmarl_edge_fill_state: too many graphs");
    hipLaunchKernelGGL(edge_fill_state_kernel, dim3((unsigned)((graphs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, offsets,
                       edge_index, edge_attr, capacity, graphs_per_env, mismatch);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_info_means(const float *info, double *means, int n_envs, int num_agents, double unreached_time, void *stream) {
    if (!info || !means || n_envs < 1 || num_agents < 1) return fail(FMARL_EINVAL, "fmarl_info_means: bad argument");
    hipLaunchKernelGGL(info_mean_kernel, dim3(FMARL_INFO_WIDTH * num_agents), dim3(256), 0, (hipStream_t)stream, info, means,
                       n_envs, num_agents, unreached_time);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

// ---- the learner's side of the rollout buffer (fmarl_learner.hip) ------------------------------------------------------
int fmarl_compute_returns(const FmarlReturns *a, const float *rewards, float *value_preds, const float *masks,
                          const float *bad_masks, const float *next_value, float *returns, void *stream) {
    if (!a || !rewards || !value_preds || !masks || !next_value || !returns) return fail(FMARL_EINVAL, "fmarl_compute_returns: null argument");
    if (a->T < 1 || a->columns < 1) return fail(FMARL_EINVAL, "fmarl_compute_returns: T and columns must be positive");
    if (a->use_proper_time_limits && !bad_masks) return fail(FMARL_EINVAL, "fmarl_compute_returns: use_proper_time_limits needs bad_masks");
    const int64_t blocks = (a->columns + 255) / 256;
    if (blocks > 0x7fffffff) return fail(FMARL_EINVAL, "fmarl_compute_returns: too many columns");
    hipLaunchKernelGGL(returns_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a, rewards, value_preds, masks, bad_masks,
                       next_value, returns);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

static_assert(sizeof(FmarlReturns) == 48 && sizeof(FmarlBatchSrc) == 144 && sizeof(FmarlBatchDst) == 136, "C-ABI layout (fair_marl_amd/_lib.py mirrors it)");
static const int kAdvBlocks = 2048;   // 8 per CU; their partials are one pass of the last block

// workspace: [mean, std (2 floats) | pad 8][sums: count, sum, sum of squares (3 doubles) | pad 8][partials: kAdvBlocks x 3 doubles]
size_t fmarl_advantages_workspace(void) { return 48 + (size_t)kAdvBlocks * 3 * sizeof(double); }

static int advantages_pass1(const float *returns, const float *value_preds, const float *active_masks, float *advantages, int64_t count,
                            int denormalize, float mean, float stddev, void *workspace, hipStream_t st, const char *who) {
    if (!returns || !value_preds || !active_masks || !advantages || !workspace || count < 1) { fail(FMARL_EINVAL, "%s: bad argument", who); return -1; }
    double *partials = (double *)((char *)workspace + 48);
    const bool wide = ((((uintptr_t)returns) | ((uintptr_t)value_preds) | ((uintptr_t)active_masks) | ((uintptr_t)advantages)) & 15) == 0;
    const int64_t want = ((wide ? count / 4 : count) + 255) / 256 + 1;
    const int blocks = (int)(want < kAdvBlocks ? want : kAdvBlocks);
    if (wide)
        hipLaunchKernelGGL(advantage_raw_kernel<4>, dim3(blocks), dim3(256), 0, st, returns, value_preds, active_masks, advantages, (size_t)count,
                           mean, stddev, denormalize ? 1 : 0, partials);
    else
        hipLaunchKernelGGL(advantage_raw_kernel<1>, dim3(blocks), dim3(256), 0, st, returns, value_preds, active_masks, advantages, (size_t)count,
                           mean, stddev, denormalize ? 1 : 0, partials);
    return blocks;
}

static int advantages_blocks(int64_t count) {
    const int64_t want = (count + 255) / 256 + 1;
    return (int)(want < kAdvBlocks ? want : kAdvBlocks);
}

int fmarl_advantages(const float *returns, const float *value_preds, const float *active_masks, float *advantages,
                     int64_t count, int denormalize, float mean, float stddev, void *workspace, void *stream) {
    const int blocks = advantages_pass1(returns, value_preds, active_masks, advantages, count, denormalize, mean, stddev, workspace,
                                        (hipStream_t)stream, "fmarl_advantages");
    if (blocks < 0) return FMARL_EINVAL;
    hipLaunchKernelGGL(advantage_scale_kernel, dim3(advantages_blocks(count)), dim3(256), 0, (hipStream_t)stream, advantages, (size_t)count,
                       (const double *)((char *)workspace + 48), blocks, (float *)workspace);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_advantages_sums(const float *returns, const float *value_preds, const float *active_masks, float *advantages,
                          int64_t count, int denormalize, float mean, float stddev, void *workspace, void *stream) {
    const int blocks = advantages_pass1(returns, value_preds, active_masks, advantages, count, denormalize, mean, stddev, workspace,
                                        (hipStream_t)stream, "fmarl_advantages_sums");
    if (blocks < 0) return FMARL_EINVAL;
    hipLaunchKernelGGL(advantage_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)((char *)workspace + 48), blocks,
                       (double *)((char *)workspace + 16));
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_advantages_apply(float *advantages, int64_t count, void *workspace, void *stream) {
    if (!advantages || !workspace || count < 1) return fail(FMARL_EINVAL, "fmarl_advantages_apply: bad argument");
    hipLaunchKernelGGL(advantage_scale_kernel, dim3(advantages_blocks(count)), dim3(256), 0, (hipStream_t)stream, advantages, (size_t)count,
                       (const double *)((char *)workspace + 16), 1, (float *)workspace);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_minibatch_gather(const FmarlBatchSrc *s, const FmarlBatchDst *d, const int64_t *index, int64_t rows, int mode,
                           int chunk, void *stream) {
    if (!s || !d || !index || rows < 0) return fail(FMARL_EINVAL, "fmarl_minibatch_gather: bad argument");
    if (rows == 0) return FMARL_OK;
    if (s->T < 1 || s->n < 1 || s->N < 1) return fail(FMARL_EINVAL, "fmarl_minibatch_gather: T, n, N must be positive");
    if (mode != 0 && mode != 1) return fail(FMARL_EINVAL, "fmarl_minibatch_gather: mode must be 0 (feed-forward) or 1 (recurrent chunks)");
    if (mode == 1 && (chunk < 1 || rows % chunk)) return fail(FMARL_EINVAL, "fmarl_minibatch_gather: rows must be a multiple of the chunk length");
    const struct { const void *dst, *src; const char *name; } need[] = {
        {d->share_obs, s->obs, "share_obs"}, {d->obs, s->obs, "obs"}, {d->node_obs, s->node_obs, "node_obs"}, {d->adj, s->adj_env, "adj"},
        {d->rnn_states, s->rnn_states, "rnn_states"}, {d->rnn_states_critic, s->rnn_states_critic, "rnn_states_critic"},
        {d->actions, s->actions, "actions"}, {d->value_preds, s->value_preds, "value_preds"}, {d->returns, s->returns, "returns"},
        {d->masks, s->masks, "masks"}, {d->active_masks, s->active_masks, "active_masks"},
        {d->old_action_log_probs, s->action_log_probs, "action_log_probs"}, {d->adv_targ, s->advantages, "advantages"},
        {d->available_actions, s->available_actions, "available_actions"}};
    for (const auto &q : need)
        if (q.dst && !q.src) return fail(FMARL_EINVAL, "fmarl_minibatch_gather: output %s needs its source array", q.name);
    const int64_t chunks = mode == 1 ? rows / chunk : 0;
    int tile = 64;   // rows per wave: fewer for a small minibatch, so that 2 048 waves (8 per CU) still have work
    while (tile > 1 && rows / tile < 2048) tile >>= 1;
    const int64_t want = (rows + 4 * tile - 1) / (4 * tile);   // four waves per workgroup
    const int blocks = (int)(want < 16384 ? want : 16384);
    hipLaunchKernelGGL(minibatch_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *s, *d, index, rows, mode, chunk, chunks, tile);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

#ifdef FMARL_MEASURE
// -DFMARL_MEASURE builds only (tools/phase_ticks.py): per-wave phase clocks of the last launch that carries FMARL_TICK sites,
// summed over the first `rows` waves into out[FMARL_TICK_PHASES] (core-clock cycles); synchronises the device.
extern "C" int fmarl_measure_ticks(double *out, int rows) {
    static unsigned int host[FMARL_TICK_ROWS][FMARL_TICK_PHASES];
    if (rows > FMARL_TICK_ROWS) rows = FMARL_TICK_ROWS;
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fmarl_ticks), sizeof(unsigned int) * FMARL_TICK_PHASES * rows));
    for (int k = 0; k < FMARL_TICK_PHASES; ++k) out[k] = 0.0;
    for (int r = 0; r < rows; ++r) for (int k = 0; k < FMARL_TICK_PHASES; ++k) out[k] += host[r][k];
    return FMARL_OK;
}
extern "C" int fmarl_measure_hstat(unsigned long long *out, int reset) {
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fmarl_hstat), sizeof(unsigned long long) * 8));
    if (reset) { unsigned long long z[8] = {}; HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_fmarl_hstat), z, sizeof z)); }
    return FMARL_OK;
}
// the raw rows: [14] / [15] = the constant-rate (100 MHz) clock at the wave's end / start
extern "C" int fmarl_measure_rows(unsigned int *out, int rows) {
    if (rows > FMARL_TICK_ROWS) rows = FMARL_TICK_ROWS;
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fmarl_ticks), sizeof(unsigned int) * FMARL_TICK_PHASES * rows));
    return FMARL_OK;
}
#endif

size_t fmarl_episode_record_words(const FmarlConfig *cfg) {
    const char *why;
    if (!config_ok(cfg, &why)) { fail(FMARL_EINVAL, "fmarl_episode_record_words: %s", why); return 0; }
    return (size_t)episode_record_words(cfg->num_agents, cfg->num_landmarks, cfg->num_obstacles, cfg->num_walls);
}

int fmarl_episode_started(void *handle) {
    Handle *h = (Handle *)handle;
    return h && h->episode_started ? 1 : 0;
}

int fmarl_pack_episode(void *handle, const void *state, void *record, void *stream) {
    Handle *h = (Handle *)handle;
    DeviceGuard on_device(h);
    if (!h || !state || !record) return fail(FMARL_EINVAL, "fmarl_pack_episode: null argument");
    Params p = bind(h, (void *)state);
    const size_t total = (size_t)p.n_envs * (p.N + p.L + p.O + p.W);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_episode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, (uint32_t *)record);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

size_t fmarl_step_record_words(const FmarlConfig *cfg) {
    const char *why;
    if (!config_ok(cfg, &why)) { fail(FMARL_EINVAL, "fmarl_step_record_words: %s", why); return 0; }
    if (cfg->scenario == FMARL_SCENARIO_FORMATION) return (size_t)kFormationRecordWords;
    if (cfg->scenario == FMARL_SCENARIO_FAIRNAV) return (size_t)(5 + 3 * cfg->num_agents);
    return 0;
}

int fmarl_rebuild_graph_rec(void *handle, const float *obs, const void *episode_record, const void *step_record, int n_envs,
                            float *node_obs, float *adj, void *stream) {
    Handle *h = (Handle *)handle;
    DeviceGuard on_device(h);
    if (!h || !episode_record || n_envs < 1 || (!node_obs && !adj)) return fail(FMARL_EINVAL, "fmarl_rebuild_graph_rec: bad argument");
    const int sc = h->cfg.scenario;
    if (sc == FMARL_SCENARIO_NAVIGATION_GRAPH ? !obs : !step_record)
        return fail(FMARL_EINVAL, "fmarl_rebuild_graph: navigation_graph needs the obs rows, the two formation scenarios the step record");
    FmarlOutputs o = {};
    o.node_obs = node_obs; o.adj = adj;
    if (!outputs_aligned(h->base, &o)) return fail(FMARL_EINVAL, "fmarl_rebuild_graph: node_obs / adj must be 16-byte aligned for this shape");
    const int grid = (n_envs + h->base.epb - 1) / h->base.epb;
    Params p = h->base;
    p.order = sc == FMARL_SCENARIO_NAVIGATION_GRAPH ? scatter_order(grid) : 1;   // (the caller's n_envs: a grid of its own)
    if (sc == FMARL_SCENARIO_FAIRNAV)
        FMARL_FNAV(h, (fairnav_rebuild_kernel<TH>), dim3(grid), h->lds_bytes, (hipStream_t)stream, p, o,
                           (const uint32_t *)episode_record, (const uint32_t *)step_record, n_envs);
    else if (sc == FMARL_SCENARIO_FORMATION)
        hipLaunchKernelGGL(formation_rebuild_kernel, dim3(grid), dim3(h->threads), h->lds_bytes, (hipStream_t)stream, p, o,
                           (const uint32_t *)episode_record, (const uint32_t *)step_record, n_envs);
    else
        hipLaunchKernelGGL(rebuild_graph_kernel, dim3(grid), dim3(kThreads), h->lds_bytes, (hipStream_t)stream, p, o, obs,
                           (const uint32_t *)episode_record, n_envs);
    HIP_OK(hipGetLastError());
    return FMARL_OK;
}

int fmarl_rebuild_graph(void *handle, const float *obs, const void *record, int n_envs, float *node_obs, float *adj,
                        void *stream) {
    return fmarl_rebuild_graph_rec(handle, obs, record, nullptr, n_envs, node_obs, adj, stream);
}

}  // extern "C"
