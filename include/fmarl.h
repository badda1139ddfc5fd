/*
 * fmarl.h -- C-ABI of the MI355X-native Fair-MARL rollout hot path (libfmarl.so).
 *
 * The reference has no FFI: its hot path sits behind a Python vectorised-env API
 * (onpolicy/envs/env_wrappers.py:850-1026 GraphSubprocVecEnv / GraphDummyVecEnv over
 * multiagent/environment.py:816-898 MultiAgentGraphEnv.step/reset).  This header is the
 * boundary a maintainer binds instead (ctypes stub: INTEGRATION.md); every entry point
 * cites the reference code it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is DEVICE memory owned by the
 *     caller (PyTorch-ROCm tensors passed as data_ptr()); the library allocates nothing
 *     persistent on the device;
 *   - every call that launches work takes the hipStream_t to launch on (as void*);
 *     calls are asynchronous and never synchronise the device (graph-capture safe);
 *   - return value 0 = ok, otherwise an FMARL_E* code; fmarl_last_error() gives the text;
 *   - one host thread per handle; handles on different devices are independent.
 */
#ifndef FMARL_H
#define FMARL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMARL_OK 0
#define FMARL_EINVAL 1   /* bad argument / unsupported shape */
#define FMARL_EHIP 2     /* a HIP call failed */

#define FMARL_SCENARIO_NAVIGATION_GRAPH 0   /* multiagent/custom_scenarios/navigation_graph.py */
#define FMARL_SCENARIO_FORMATION 1          /* multiagent/custom_scenarios/fair_graph_formation.py */
#define FMARL_SCENARIO_FAIRNAV 2            /* multiagent/custom_scenarios/nav_fairassign_fairrew_formation_graph.py */

/* FmarlConfig.flags.  ASYNC_RESET: the next episode's placement + fair assignment (a pure function of
 * seed, env and episode index) is computed ahead of time on a library-owned side stream while the current
 * episode runs; a reset then only commits the staged state and emits the observation.  The staging launches are
 * enqueued by the first fmarl_step call AFTER a reset (not by the reset itself: they belong to a later episode and
 * would otherwise sit behind the caller's synchronisation on the step that ended this one).  navigation_graph with all envs
 * in lockstep: the fmarl_step call that ends an episode commits the staged episode and emits its first observation in the same
 * launch (step_end_kernel) instead of launching a commit and an emission kernel behind a step that emits nothing.  Results are
 * identical to the synchronous path.  hipGraph capture (see fmarl_step): with FMARL_RESET_LOCKSTEP the side stream becomes a
 * forked branch of the caller's graph -- capture whole episodes, starting with the first step after a reset, so that every
 * staging is joined by the episode end inside the graph; with FMARL_RESET_AUTO the handle turns into a synchronous-reset
 * handle for good (the device-checked resets of such a graph are the synchronous launches).  fmarl_reset itself is not
 * capturable on such a handle. */
#define FMARL_FLAG_ASYNC_RESET 1
/* GLOBAL_FEATURES: --graph_feat_type global (navigation_graph.py:981-1009, 1058-1077): node_obs rows are the
 * 7 absolute columns [vel, pos, goal, type], identical for every ego agent (F = 7).  navigation_graph without
 * walls only: the reference's _get_entity_feat_global raises for walls and the other two scenarios are not built. */
#define FMARL_FLAG_GLOBAL_FEATURES 2

/* Scenario arguments: multiagent/custom_scenarios/navigation_graph.py:94-129,208
 * (defaults onpolicy/config.py:176-252, onpolicy/scripts/train_mpe.py:71-106). */
typedef struct FmarlConfig {
    int32_t scenario;        /* FMARL_SCENARIO_* */
    int32_t n_envs;          /* environments in this handle (n_rollout_threads of this shard) */
    int32_t num_agents;      /* N.  Limits the reference does not have (its Python loops take any N): navigation_graph 1..64 (the fair
                              * assignment runs on one 64-lane wave per env); fair_graph_formation 1..32 and
                              * nav_fairassign_fairrew_formation_graph 2..32 (32-bit occupancy masks, one env inside one wave).
                              * fmarl_create / fmarl_state_bytes refuse other values with FMARL_EINVAL and a text naming the range;
                              * the reference's own experiments use 3..10 agents (onpolicy/scripts/train_mpe.py:71-106). */
    int32_t num_landmarks;   /* L (navigation_graph needs L == N for the assignment) */
    int32_t num_obstacles;   /* O */
    int32_t num_walls;       /* W (<= 2: navigation_graph.py:289 has two wall axes) */
    int32_t episode_length;  /* done when current_step >= episode_length (environment.py:237-247) */
    int32_t has_max_speed;   /* 0 = max_speed None */
    int32_t env_offset;      /* global index of env 0 (RNG streams independent of the sharding) */
    int32_t flags;           /* FMARL_FLAG_* */
    double world_size;
    double max_speed;
    double collision_rew;
    double goal_rew;
    double min_dist_thresh;
    double fair_rew;
    double zeroshift;
    double max_edge_dist;
    double min_obs_dist;     /* FAIRNAV only (onpolicy/config.py:188) */
    uint64_t seed;           /* Philox key; env e, episode k draw from counter (i, env_offset+e, k, TAG) */
    int32_t envs_per_workgroup; /* launch geometry: 0 = the library's choice (as many envs per 256-thread workgroup as LDS
                                 * allows, fewer for batches too small to fill the chip otherwise); k > 0 = at most k, whatever
                                 * the batch size (fair_graph_formation: rounded up to a multiple of 4, one env per wave at
                                 * least).  Results never depend on it -- the parity tests run every fixture at 1, 2 and the
                                 * full-batch geometry; fmarl_envs_per_workgroup() reports what a handle uses. */
    int32_t reserved0;       /* 0 */
} FmarlConfig;

/* Output buffers of one step / reset, all caller-owned device memory, float32 unless noted.
 * Shapes follow onpolicy/envs/env_wrappers.py:988-1002 with two differences that the Python
 * layer undoes in its NumPy-compat mode: adj is stored once per env (the reference returns
  * the same E x E matrix N times, navigation_graph.py:1033) and infos are dense field-major
 * record planes instead of per-agent dicts (navigation_graph.py:625-647).  Any pointer may be NULL
 * to skip that output.  node_obs and adj must be 16-byte aligned when their rows are multiples of 16 bytes
 * (E * F % 4 == 0 resp. E % 4 == 0: written with 16-byte stores; hipMalloc'd memory always is); other shapes
 * take any float pointer.  Misaligned buffers are refused with FMARL_EINVAL. */
typedef struct FmarlOutputs {
    float *obs;          /* (n, N, D)        D = 7 navigation_graph / 6 formation / 11 fairnav */
    float *node_obs;     /* (n, N, E, F)     F = 11 / 12 / 13, E = N + L + O + W              */
    float *adj;          /* (n, E, E)        cached_dist_mag, multiagent/core.py:204-228      */
    float *reward;       /* (n, N)                                                            */
    uint8_t *done;       /* (n, N)           environment.py:237-247                           */
    float *info;         /* (FMARL_INFO_WIDTH, n, N) field-major records, FMARL_INFO_* order   */
    int32_t *edge_nnz;   /* (n)              policy edges of the env's graph: entries of adj with 0 < d < max_edge_dist
                                              (onpolicy/algorithms/utils/gnn.py:307-326 processAdj), counted by the adj
                                              emission itself; needs adj.  See fmarl_edge_offsets / fmarl_edge_fill_state */
    uint32_t *graph_record; /* (n, N, fmarl_step_record_words) per-step part of the cross-GPU graph hand-off of the
                                              scenarios whose node features depend on per-step scenario state
                                              (the two formation scenarios); see fmarl_rebuild_graph_rec                   */
} FmarlOutputs;

#define FMARL_INFO_WIDTH 14
/* order of the info record (keys of navigation_graph.py:625-647 + environment.py:861) */
enum {
    FMARL_INFO_DIST_TO_GOAL = 0, FMARL_INFO_TIME_REQ_TO_GOAL, FMARL_INFO_NUM_AGENT_COLLISIONS,
    FMARL_INFO_NUM_OBST_COLLISIONS, FMARL_INFO_DISTANCE_MEAN, FMARL_INFO_DISTANCE_VARIANCE,
    FMARL_INFO_MEAN_BY_VARIANCE, FMARL_INFO_DISTS_TRAVELED, FMARL_INFO_TIME_TAKEN,
    FMARL_INFO_TIME_MEAN, FMARL_INFO_TIME_STDDEV, FMARL_INFO_TIME_MEAN_BY_STDDEV,
    FMARL_INFO_MIN_TIME_TO_GOAL, FMARL_INFO_INDIVIDUAL_REWARD
};
/* fair_graph_formation has no time statistics; slot 9 carries its 'Formation_dist' key and slots
 * 8, 10, 11 are zero (fair_graph_formation.py:484-499). */
#define FMARL_INFO_FORMATION_DIST FMARL_INFO_TIME_MEAN

/* World state: ONE caller-owned device buffer of fmarl_state_bytes() bytes holding the fields
 * below back to back (each 256-byte aligned, row-major, env-major), i.e. the SoA restatement of
 * multiagent/core.py:12-19,60-128 EntityState/Entity/Agent/Wall objects and the per-world
 * vectors of navigation_graph.py:212-225.  float64 where the reference holds real values in float64; integer-valued
 * reference fields (counters, indices, flags) as i32 / i8 -- fmarl_get_state / fmarl_set_state copy the field in the dtype
 * fmarl_state_field reports. */
enum {
    FMARL_F_AGENT_POS = 0,     /* f64 (n, N, 2)  agent.state.p_pos                         */
    FMARL_F_AGENT_VEL,         /* f64 (n, N, 2)  agent.state.p_vel                         */
    FMARL_F_P_DIST,            /* f64 (n, N)     agent.state.p_dist                        */
    FMARL_F_LANDMARK_POS,      /* f64 (n, L, 2)                                            */
    FMARL_F_OBSTACLE_POS,      /* f64 (n, O, 2)                                            */
    FMARL_F_WALL_AXIS,         /* f64 (n, W)     wall.axis_pos                             */
    FMARL_F_WALL_E0,           /* f64 (n, W)     wall.endpoints[0]                         */
    FMARL_F_WALL_E1,           /* f64 (n, W)     wall.endpoints[1]                         */
    FMARL_F_WALL_ORIENT,       /* i32 (n, W)     0 = 'H', 1 = 'V'                          */
    FMARL_F_WALL_LENGTH,       /* f64 (n)        scenario.wall_length (drawn in make_world) */
    FMARL_F_GOAL_MATCH,        /* i32 (n, N)     scenario.goal_match_index                 */
    FMARL_F_DISTS_TO_GOAL,     /* f64 (n, N)     world.dists_to_goal                       */
    FMARL_F_TIMES_REQUIRED,    /* f64 (n, N)     world.times_required                      */
    FMARL_F_DIST_LEFT,         /* f64 (n, N)     world.dist_left_to_goal                   */
    FMARL_F_NUM_OBST_COLL,     /* i32 (n, N)     world.num_obstacle_collisions             */
    FMARL_F_NUM_AGENT_COLL,    /* i32 (n, N)     world.num_agent_collisions                */
    FMARL_F_MIN_TIME,          /* f64 (n, N)     agent.goal_min_time                       */
    FMARL_F_CUR_STEP,          /* i32 (n)        env.current_step == world.current_time_step */
    FMARL_F_EPISODE,           /* i32 (n)        resets drawn so far (Philox counter word)  */
    FMARL_F_SLOT_POS,          /* f64 (n, N, 2)  formation: scenario.expected_poses         */
    FMARL_F_SLOT_OCC,          /* f64 (n, N)     formation: scenario.expected_poses_occupied */
    FMARL_F_SLOT_DELTA,        /* f64 (n, N)     formation: scenario.delta_dists            */
    FMARL_F_FORMATION_DONE,    /* f64 (n, N)     formation: world.formation_complete        */
    FMARL_F_GOAL_OCC,          /* f64 (n, N)     fairnav: scenario.landmark_poses_occupied (real-valued: 0, 1, a distance, 1 - a distance) */
    FMARL_F_GOAL_HISTORY,      /* i8  (n, N)     fairnav: scenario.goal_history (reference: float array holding -1 or an agent index) */
    FMARL_F_GOAL_REACHED,      /* i8  (n, N)     fairnav: scenario.goal_reached (reference: float array holding -1 or a goal index)   */
    FMARL_F_STATUS,            /* i8  (n, N)     fairnav: agent.status (reference: Python bool)                                       */
    FMARL_F_RESET_FLAG,        /* i32 (n)        internal: envs picked by the last reset launch */
    FMARL_F_STAGE_AGENT_POS,   /* f64 (n, N, 2)  staged next episode (FMARL_FLAG_ASYNC_RESET) ...          */
    FMARL_F_STAGE_LANDMARK_POS,/* f64 (n, L, 2)                                                            */
    FMARL_F_STAGE_OBSTACLE_POS,/* f64 (n, O, 2)                                                            */
    FMARL_F_STAGE_WALL_AXIS,   /* f64 (n, W)                                                               */
    FMARL_F_STAGE_WALL_ORIENT, /* i32 (n, W)                                                               */
    FMARL_F_STAGE_GOAL_MATCH,  /* i32 (n, N)     ... including its fair assignment                         */
    FMARL_F_STAGE_VALID,       /* i32 (n)        1 = the staged data belongs to episode index `episode`    */
    FMARL_F_STAGE_NEED,        /* i32 (n)        internal: envs being staged                               */
    FMARL_F_PLACE_FAILS,       /* i32 (n)        entities of the env's current episode that were placed although all
                                                 10 000 rejection-sampling draws collided (the reference would loop forever,
                                                 navigation_graph.py:389-457, :472-535): 0 unless the world is over-crowded */
    FMARL_F_STAGE_PLACE_FAILS, /* i32 (n)        the same for the staged next episode (FMARL_FLAG_ASYNC_RESET)             */
    FMARL_F_MATCH_DUAL,        /* f64 (n, N)     internal (formation): column potentials of the last slot matching, the
                                                 warm start of the next one (any values are valid: the optimum is unique) */
    FMARL_F_ROT_TABLE,         /* f64 (N, 2)     internal (formation), ONE table for all envs: (cos, sin) of i * 2 pi / N -- the slots on
                                                 the circle are the anchor direction rotated by these (fair_graph_formation.py:630-648).
                                                 A copy only: the kernels read the handle's own table (512 bytes of device memory
                                                 allocated by fmarl_create), so a caller may zero, restore or copy this field freely */
    FMARL_NUM_FIELDS
};
#define FMARL_DTYPE_F64 0
#define FMARL_DTYPE_I32 1
#define FMARL_DTYPE_I8 2    /* small integer-valued fields of the reference (indices, flags): one byte instead of a float64 */

/* --- lifetime ---------------------------------------------------------------------------- */

/* Replaces GraphMPEEnv(args) x n_envs (multiagent/MPE_env.py:55-77): validates the config and
 * builds the host-side handle (launch geometry, constants).  No device memory is allocated. */
int fmarl_create(const FmarlConfig *cfg, void **handle);
int fmarl_destroy(void *handle);
const char *fmarl_last_error(void);
/* Envs per workgroup of the handle's step / reset-emission launches (see FmarlConfig.envs_per_workgroup). */
int fmarl_envs_per_workgroup(void *handle);

/* Size and layout of the state buffer for cfg (0 / FMARL_EINVAL on a bad config). */
size_t fmarl_state_bytes(const FmarlConfig *cfg);
int fmarl_state_field(const FmarlConfig *cfg, int field, size_t *offset_bytes,
                      size_t *count, int *dtype);

/* --- the hot path ------------------------------------------------------------------------ */

/* make_world for every env (navigation_graph.py:48-210): default vectors, wall_length draw and
 * the hidden first reset_world (:209).  Must be called once on a fresh state buffer. */
int fmarl_init_state(void *handle, void *state, void *stream);

/* MultiAgentGraphEnv.reset (environment.py:882-898) = scenario.reset_world + random_scenario
 * (navigation_graph.py:212-575) incl. the fair goal assignment (marl_fair_assign.py:16-55), for the
 * envs with env_mask[e] != 0 (NULL = all), then obs / node_obs / adj of those envs into outs. */
int fmarl_reset(void *handle, void *state, const uint8_t *env_mask,
                const FmarlOutputs *outs, void *stream);

/* MultiAgentGraphEnv.step (environment.py:816-877): action decode (:265-311), World.step
 * (multiagent/core.py:250-274: action force, O(E^2) entity + wall collision forces, integrate),
 * calculate_distances (core.py:204-228), then per agent observation / reward / graph_observation
 * / done / info_callback (navigation_graph.py:826-857, 760-824, 941-1035, 577-647).
 * Exactly one of action_idx (int32 (n, N), value 0..4) and action_vec (float32 (n, N, 5), the
 * reference's one-hot / continuous form) is non-NULL.
 * auto_reset != 0 adds the vec-env worker's behaviour (env_wrappers.py:859-865): envs whose agents
 * are all done are reset and their obs / node_obs / adj are the reset observation while
 * reward / done / info stay those of the terminal step.
 * hipGraph capture: the call only enqueues work on `stream` (and, with FMARL_FLAG_ASYNC_RESET + FMARL_RESET_LOCKSTEP, on the
 * handle's side stream forked from it and joined back), so a run of steps can be captured.  With auto_reset = FMARL_RESET_AUTO a captured step always enqueues the auto-reset launches,
 * which test every env's step counter on the device -- a graph may hold any number of steps and be replayed from any
 * phase of an episode; the handle stops mirroring the step counter on the host from then on.
 * auto_reset = FMARL_RESET_LOCKSTEP is the lean variant for launch-bound batches: while all envs share one step counter
 * (fmarl_get_phase() >= 0) a captured step takes the reset-or-not decision from the host's mirror, exactly like an eager
 * step, so the graph holds one reset per episode instead of one test per step.  Such a graph is valid only when
 * replayed from the episode phase it was captured at: after capturing T steps call fmarl_set_phase(handle, phase before
 * the capture) (nothing ran), and after every replay fmarl_set_phase(handle, (phase + T) % episode_length).  Outside
 * capture the two values behave the same. */
#define FMARL_RESET_AUTO 1
#define FMARL_RESET_LOCKSTEP 2
int fmarl_step(void *handle, void *state, const int32_t *action_idx, const float *action_vec,
               const FmarlOutputs *outs, int auto_reset, void *stream);

/* A run of n_steps consecutive steps from an action tape in as few launches as possible (action indices; any scenario, any
 * state of the handle): equivalent to n_steps fmarl_step(..., FMARL_RESET_AUTO ...) calls where step t reads action_idx + t *
 * span->actions and writes the outputs of `outs` shifted by the span's per-step strides (in elements; 0 = every step writes the
 * same buffer, e.g. a rollout that only needs the last observation; the slots of a rollout buffer laid out (T, n, ...) have
 * stride n * ...).  Envs never interact, so inside a span every workgroup walks its own envs through the steps without waiting
 * for the rest of the batch: no per-step launch, no per-step head and tail of the grid.  The step that ends an episode is a
 * launch of its own (it commits / resets and, with the staged reset, waits for the staging that ran beside the span); so is
 * every step while the envs are not in lockstep.  nav_fairassign_fairrew_formation_graph, whose envs end their episodes one
 * by one and are reset inside the step, runs ALL n_steps as one launch (fairnav_span_kernel: the state goes through L2
 * between the steps; 65 536 x 3: 0.059 -> 0.050 ms per step, profiles/r4_notes.md).
 * The scripted / random-action rollout of the reference's throughput runs; a policy in the loop needs fmarl_step.  Not
 * capturable into a hipGraph (it decides on the host where episodes end; it needs no graph: an episode is three launches).
 * Measured (profiles/archive/r3_notes.md): 10 agents x 65 536 envs 0.250 -> 0.199 ms per step, 3 agents x 4 096 envs 14.5 -> 11.5 us. */
typedef struct FmarlSpan {
    int64_t obs, node_obs, adj, reward, done, info, edge_nnz, graph_record;   /* per-step strides of the outputs, in elements */
    int64_t actions;   /* per-step stride of the action tape in elements: n * N for dense action_idx, n * N * 5 for action_vec */
} FmarlSpan;
/* Exactly one of action_idx (T, n, N) int32 / action_vec (T, n, N, 5) float32, as for fmarl_step. */
int fmarl_step_span(void *handle, void *state, const int32_t *action_idx, const float *action_vec, int n_steps,
                    const FmarlOutputs *outs, const FmarlSpan *span, void *stream);

/* Host-side mirror of the envs' common step counter: steps since the last reset of all envs, or -1 when the envs are not
 * known to be in lockstep (masked resets, fmarl_set_state, graphs captured with FMARL_RESET_AUTO, fairnav).  No device access. */
int fmarl_get_phase(void *handle);
int fmarl_set_phase(void *handle, int phase);

/* Copy one state field (FMARL_F_*) out of / into the state buffer, in the field's own shape and dtype
 * (= the reference's attribute layout, e.g. agent.state.p_pos as f64 (n, N, 2)).  `dst` / `src` may be host
 * or device memory (hipMemcpyDefault).  fmarl_set_state implies fmarl_state_changed.  This is the parity
 * harness' way to inject / read worlds (SURVEY.md App. C does the same on the reference's objects). */
int fmarl_get_state(void *handle, const void *state, int field, void *dst, void *stream);
int fmarl_set_state(void *handle, void *state, int field, const void *src, void *stream);

/* Tell the handle the caller wrote into the state buffer (parity harness set_state): drops the
 * host-side "all envs share one step counter" shortcut used to skip the auto-reset launch. */
int fmarl_state_changed(void *handle);

/* Measurement hook (bench.py): record a hipEvent pair around every step-kernel launch of this
 * handle, on the launch stream.  enable(capacity) (re)starts recording into `capacity` pairs
 * (0 = off); read() returns the per-launch durations [ms] recorded since then and (steps may be NULL) how many env
 * steps each launch covered (1 for fmarl_step, the run length for a span launch) -- the caller
 * must have synchronised the stream -- and restarts. */
int fmarl_profile_enable(void *handle, int capacity);
int fmarl_profile_read(void *handle, float *ms, int *steps, int max_count, int *count);
/* What fmarl_step has enqueued on this handle so far (host-side counters, no device access): counts[0] step-kernel launches,
 * counts[1] of those that also committed the staged episode and emitted its first observation (the folded episode end,
 * FMARL_FLAG_ASYNC_RESET above), counts[2] step calls followed by separate auto-reset launches, counts[3] stagings of a next
 * episode on the side stream.  bench.py derives the bytes a launch writes from these instead of from the configuration. */
int fmarl_launch_counts(void *handle, int64_t *counts);
/* The launch geometry of the step kernels of this handle: geometry[0] workgroups, [1] threads per workgroup, [2] dynamic LDS
 * bytes per workgroup, [3] envs per workgroup.  (Residency: 160 KB of LDS per CU; a workgroup of exactly 40 960 bytes was
 * measured to fit only three times, 40 320 four times.) */
int fmarl_launch_geometry(void *handle, int64_t *geometry);

/* Test hook (tests/test_hip_parity.py): fill the LDS of every CU with 0xFF bytes (one launch of workgroups that take
 * 64 KB each and write all of it).  LDS is not cleared between kernels, so a table a step kernel reads before writing
 * holds whatever the previous kernel left there: usually its own earlier values (the bug stays invisible), after this
 * call a NaN pattern / all-ones words (it shows at once). */
int fmarl_poison_lds(void *handle, void *stream);

/* Measurement aid (bench.py `store_ceiling_ms`): a kernel that does nothing but write `bytes` bytes to `dst` with 16-byte
 * stores, in the shapes the emission writes in -- shape 0: flat grid-stride stream; 1: a workgroup streams a contiguous chunk of
 * `chunk_bytes`; 2: every wave streams its own contiguous quarter of such a chunk (1 KiB per store instruction); 3 / 4: shapes
 * 1 / 2 with non-temporal stores.  Workgroup b
 * writes chunk (b * order) mod n_chunks (order 1 = dispatch order; a large order coprime with the number of chunks scatters the
 * resident workgroups over the buffer); persist > 0: that many workgroups live for the whole launch and take chunks round-robin
 * (a span's long-lived workgroups), 0: one workgroup per chunk.  The best of them over the byte count of a step is the box's
 * write ceiling for that step: no step kernel can be faster than its own store stream.  Replaces nothing in the reference
 * (which has no device path); takes no handle. */
int fmarl_store_stream(void *dst, size_t bytes, int shape, size_t chunk_bytes, int order, int persist, void *stream);

/* Measurement aid, as fmarl_store_stream: the store stream of the GENERIC emission path -- node rows whose width is not a multiple of
 * 16 bytes (10 agents, E = 23: 1 012-byte ego rows) leave a wave as windows of `window_bytes` (64 rows) at 4-byte aligned starts,
 * aligned 16-byte chunks per lane plus up to three dwords at either end, and the adjacency as one dword per lane and store -- with
 * everything but the stores removed.  Workgroup b owns group (b * order) mod groups of every time slot: node_group_bytes of node[] and
 * adj_group_bytes of adj[] (the last group of a slot is cut at node_slot_bytes / adj_slot_bytes), and walks the `slots` slots in order
 * like a span launch.  Its time per slot is the ceiling of a step kernel that writes in this pattern.  Takes no handle. */
int fmarl_store_pattern(void *node, void *adj, size_t node_group_bytes, size_t adj_group_bytes, int groups, int slots, size_t node_slot_bytes,
                        size_t adj_slot_bytes, int window_bytes, int order, void *stream);

/* The masks of the runner's insert (onpolicy/runner/shared/graph_mpe_runner.py:444-465) for `rows` env-steps of num_agents agents each:
 * done u8 (rows, N) -> masks f32 (rows, N): 0 where the agent is done; active_masks f32 (rows, N): 0 where the agent is done but
 * its env is not.  One launch (DeviceRolloutBuffer.insert_step / insert_span). */
int fmarl_insert_masks(const uint8_t *done, float *masks, float *active_masks, int64_t rows, int num_agents, void *stream);

/* Memory for time slots (fair_marl_amd.OutputRing; the layout of the reference's rollout storage, onpolicy/utils/graph_buffer.py:84-110:
 * a (slots, n, ...) array, slot t = the bytes [t * slot_bytes, (t + 1) * slot_bytes) behind *base).  The slots are virtually
 * contiguous, but their PHYSICAL memory is interleaved: the array is backed by slots * slot_bytes / piece_bytes physical pieces
 * (hipMemCreate), and virtual piece j of slot t is mapped to physical piece j * slots + t -- every slot is spread evenly over the
 * whole allocation.  Why: MI355X takes a store stream at 5.7-6.0 TB/s when its target is one contiguous 8 GB region and at 6.8-7.1
 * TB/s when the same bytes are spread over 160 GB, even in pieces of 32 MiB (profiles/r4_spread_probe.txt, profiles/r4_notes.md) -- a
 * launch that writes ONE time slot (a policy in the loop: fmarl_step) gets the rate of the whole ring.  piece_bytes = 0: the
 * library's choice (<= 16 MiB, a divisor of slot_bytes); otherwise a multiple of the allocation granularity that divides
 * slot_bytes.  FMARL_EINVAL when the slot size has no such divisor, FMARL_EHIP when the device has no virtual memory management:
 * allocate plainly then.  The memory belongs to the caller until fmarl_ring_free(cookie) (no launch may still use it).
 * fmarl_ring_free returns the physical memory; the array's virtual address range stays reserved, idle, for the life of the process:
 * on this stack the GPU holds on to translations of unmapped addresses, whether the range goes back to the runtime and is reserved
 * again (round 4) or stays with the process and gets fresh pieces mapped into it (round 5: tried, same fault) -- an address that
 * has carried a mapping is never used again (tools/vmm_fault_repro.cpp shows both faults on the bare HIP calls).  The reservations
 * are counted and capped (8 TiB per process; FMARL_RING_RESERVE_CAP_GB overrides): past the cap FMARL_EINVAL -- allocate plainly.
 * An array allocated AFTER an earlier one of the process was freed (the only condition a fault was ever seen under) is checked before
 * it is handed out: a kernel fills it, a kernel reads it back and leaves zeroes; if a word is wrong the array is released, the event
 * logged to stderr and FMARL_EHIP returned -- allocate plainly (FMARL_RING_VERIFY=1 checks every array, =0 none).  fmarl_ring_free
 * waits for the device itself (hipDeviceSynchronize) before it unmaps.
 * Access is granted to the allocating device; with FMARL_RING_PEER_ACCESS=1 in the environment also to every device that has peer
 * access to it (hipMalloc memory is peer-accessible once peer access is enabled, an array of pieces only for the devices named). */
int fmarl_ring_alloc(size_t slot_bytes, int slots, size_t piece_bytes, void **base, void **cookie);
int fmarl_ring_free(void *cookie);
/* The allocator's books: out[0] bytes of address space reserved so far, [1] of them idle (ranges of freed arrays), [2] ranges
 * reserved, [3] of them idle, [4] the cap on [0] in bytes, [5] requests refused at the cap, [6] arrays checked by a kernel's fill
 * before they were handed out, [7] of them refused because they did not hold it. */
int fmarl_ring_stats(uint64_t out[8]);

/* --- pieces exported on their own -------------------------------------------------------- */

/* cdist(agent_pos, goal_pos) of navigation_graph.py:555: f64 (n, N, 2) x (n, L, 2) -> (n, N, L). */
int fmarl_cost_matrix(const double *agent_pos, const double *goal_pos, double *costs,
                      int n_envs, int num_agents, int num_goals, void *stream);

/* solve_fair_assignment (marl_fair_assign.py:16-55): lexicographic-bottleneck assignment of
 * costs f64 (n, N, N) -> perm i32 (n, N), perm[e][i] = goal of agent i (N <= 64). */
int fmarl_lexifair(const double *costs, int32_t *perm, int n_envs, int num_agents, void *stream);

/* Scenario.update_graph (navigation_graph.py:1037-1056): COO edges with 0 < adj <= max_edge_dist in
 * row-major order.  adj f32 (n, E, E) -> edge_index i32 (n, 2, E*E) (padded with -1),
 * edge_weight f32 (n, E*E), nnz i32 (n). */
int fmarl_update_graph(const float *adj, int32_t *edge_index, float *edge_weight, int32_t *nnz,
                       int n_envs, int num_entities, double max_edge_dist, void *stream);

/* The same from the world state, exactly as the reference computes it: float64 distances of the entities
 * (multiagent/core.py:204-228, np.linalg.norm), `<=` against the float64 cfg.max_edge_dist, float64 weights.  The
 * float32 adj output cannot decide a distance within a float32 ulp of the threshold; this entry point can.  Call it
 * where MultiAgentGraphEnv.step does (environment.py:817-818: before the step).
 * -> edge_index i32 (n, 2, E*E) (padded with -1), edge_weight f64 (n, E*E), nnz i32 (n). */
int fmarl_update_graph_state(void *handle, const void *state, int32_t *edge_index, double *edge_weight, int32_t *nnz,
                             void *stream);

/* Policy-side edge construction, onpolicy/algorithms/utils/gnn.py:307-326 processAdj (strict != 0:
 * 0 < adj < max_edge_dist; strict == 0: the <= of update_graph) with the node-id offsets of the PyG batch
 * of :243-253.  Pass 1: nnz i32 (n) per env.  The caller builds offsets i64 (n_graphs + 1) = exclusive
 * prefix sum of the per-graph counts (graph b uses env b / graphs_per_env; the reference replicates the
 * matrix per agent).  Pass 2: edge_index i64 (2, total) row-major [rows | cols] with ids b * E + r,
 * edge_attr f32 (total). */
int fmarl_edge_count(const float *adj, int32_t *nnz, int n_envs, int num_entities, double max_edge_dist,
                     int strict, void *stream);
int fmarl_edge_fill(const float *adj, const int64_t *offsets, int64_t *edge_index, float *edge_attr, int64_t total,
                    int n_graphs, int graphs_per_env, int num_entities, double max_edge_dist, int strict, void *stream);

/* processAdj fused with the step: the adj emission of fmarl_step / fmarl_reset counts every env's
 * policy edges into FmarlOutputs.edge_nnz; the prefix sum and the edge list are built on the device from the WORLD STATE
 * (the float32 entity positions adj was computed from), so adj is never read back and nothing synchronises with the host.
 *   fmarl_edge_offsets: offsets i64 (n_envs * graphs_per_env + 1) = exclusive prefix sum of the per-graph counts
 *       (graph b belongs to env b / graphs_per_env); offsets[last] = total number of edges.
 *   fmarl_edge_fill_state: edge_index i64 (2, capacity) rows | cols with node ids b * E + r, edge_attr f32 (capacity), in
 *       the row-major order of processAdj; edges beyond `capacity` are dropped (size the buffers with offsets[last], or
 *       with an upper bound when the host must not wait).  Call it after the fmarl_step / fmarl_reset whose adj it
 *       describes and before the next step.  edge_attr equals the adj entries bit for bit.  The offsets must come from the
 *       edge_nnz of THAT step's output set: a graph only ever writes inside [offsets[b], offsets[b + 1]), and `mismatch`
 *       (device int32 counter, may be NULL) is incremented for every graph whose edges recomputed from the state do not
 *       fill that range exactly (counts of another output set, a state rewritten in between). */
int fmarl_edge_offsets(const int32_t *nnz, int n_envs, int graphs_per_env, int64_t *offsets, void *stream);
int fmarl_edge_fill_state(void *handle, const void *state, const int64_t *offsets, int64_t *edge_index, float *edge_attr,
                          int64_t capacity, int graphs_per_env, int32_t *mismatch, void *stream);

/* Cross-GPU hand-off of the graph observation (navigation_graph; fair_graph_formation: see fmarl_step_record_words).  The reference's workers pipe node_obs /
 * adj to the learner with every step (onpolicy/envs/env_wrappers.py:988-996).  Between GPUs only the compact
 * record travels: the per-step obs rows, which carry every agent's velocity and position
 * (navigation_graph.py:855-857), and once per episode the entities World.step never moves -- each agent's goal,
 * the landmarks, obstacles and walls placed by reset_world (navigation_graph.py:264-575).
 *   fmarl_episode_record_words: 32-bit words per env of the episode record:
 *       goal (x, y) f32 x N | landmark, obstacle (x, y) f32 x (L + O) | wall [axis f64, e0 f32, e1 f32, orient f32, 0] x W
 *   fmarl_episode_started: 1 if the last fmarl_reset / fmarl_step call may have started episodes (host-side
 *       knowledge: always 1 after a reset, after a step once per episode_length steps while all envs run in
 *       lockstep, after every auto-resetting step otherwise); no device access.
 *   fmarl_pack_episode: record (n_envs, words) <- state.
 *   fmarl_rebuild_graph: (obs f32 (n_envs, N, D), record (n_envs, words)) -> node_obs f32 (n_envs, N, E, F) and / or
 *       adj f32 (n_envs, E, E) as graph_observation would give them (navigation_graph.py:941-1035, 1079-1124);
 *       n_envs is the caller's (e.g. all ranks' envs), the handle supplies the entity counts.  node_obs equals the
 *       sender's bit for bit; adj is computed from the f32 positions (difference < 1e-6). */
size_t fmarl_episode_record_words(const FmarlConfig *cfg);
/* fair_graph_formation: the node features also depend on what the scenario's sequential agent loop left behind in this
 * step (slots on the circle, the per-ego occupancy / goal-branch masks, fair_graph_formation.py:810-971), which the obs
 * rows do not carry (obs = concat(v, x, goal - x) + flag, :740-741).  The step kernel therefore writes a compact record
 * beside obs -- 32-bit words per agent: x, y, vx, vy, slot x, slot y (f32), branch mask, flag mask, nearest slot | matched
 * slot << 8 -- 36 B per agent-step against 768 B of node_obs at BASELINE config 4; fmarl_rebuild_graph_rec expands it
 * with the same emission code.  nav_fairassign_fairrew_formation_graph (occupancy / history walk,
 * nav_fairassign_fairrew_formation_graph.py:1222-1334): 5 + 3 N words per agent -- x, y, vx, vy, newly-stopped (f32), then
 * for every agent entity of the agent's row block (goal landmark index or -1, occupancy, history).  0 words for
 * navigation_graph (obs + the episode record are enough). */
size_t fmarl_step_record_words(const FmarlConfig *cfg);
int fmarl_rebuild_graph_rec(void *handle, const float *obs, const void *episode_record, const void *step_record, int n_envs,
                            float *node_obs, float *adj, void *stream);
int fmarl_episode_started(void *handle);
int fmarl_pack_episode(void *handle, const void *state, void *record, void *stream);
int fmarl_rebuild_graph(void *handle, const float *obs, const void *record, int n_envs, float *node_obs, float *adj,
                        void *stream);

/* Per-agent means over the envs of every info field = what process_infos + log_env report
 * (onpolicy/runner/shared/base_runner.py:197-306); Time_req_to_goal == -1 counts as unreached_time
 * (= episode_length * dt, :212-215).  info f32 (FMARL_INFO_WIDTH, n, N) -> means f64 (FMARL_INFO_WIDTH, N). */
int fmarl_info_means(const float *info, double *means, int n_envs, int num_agents, double unreached_time, void *stream);

/* ---- the learner's side of the rollout buffer ---------------------------------------------------
 * What the reference's trainer computes with NumPy on the filled GraphReplayBuffer before its first gradient step.  The
 * arrays are the buffer's own, C-contiguous float32 with the trailing 1 of (T, n, N, 1) dropped: `columns` = n * N.
 * Stateless; they run on the current device.
 *
 * fmarl_compute_returns = GraphReplayBuffer.compute_returns (onpolicy/utils/graph_buffer.py:285-366), every branch:
 * use_gae x use_proper_time_limits x value normaliser (ValueNorm / PopArt: denormalize(x) = x * stddev + mean in float32,
 * onpolicy/utils/valuenorm.py:92-104, onpolicy/algorithms/utils/popart.py:101-111; the caller passes the normaliser's
 * current debiased mean and sqrt(var)).  rewards (T, columns); value_preds, masks, bad_masks, returns (T + 1, columns);
 * next_value (columns).  With use_gae, value_preds[T] = next_value is written as the reference does and returns[T] is left
 * alone; without, returns[T] = next_value.  bad_masks may be NULL unless use_proper_time_limits.  Float32 in the
 * reference's order of operations: equal to NumPy's result bit for bit. */
typedef struct FmarlReturns {
    double gamma, gae_lambda;      /* args.gamma, args.gae_lambda (Python floats: their product is formed in double) */
    float mean, stddev;            /* of the value normaliser; ignored unless denormalize */
    int32_t denormalize, use_gae, use_proper_time_limits;
    int32_t T;                     /* episode_length = rewards.shape[0] */
    int64_t columns;               /* n_rollout_threads * num_agents */
} FmarlReturns;
int fmarl_compute_returns(const FmarlReturns *args, const float *rewards, float *value_preds, const float *masks,
                          const float *bad_masks, const float *next_value, float *returns, void *stream);

/* The advantages GR_MAPPO.train hands to the generators (onpolicy/algorithms/graph_mappo.py:294-304):
 * returns[:T] - denormalize(value_preds[:T]), standardised by the mean / standard deviation over the entries whose active
 * mask is not 0, (x - mean) / (std + 1e-5).  count = T * columns entries of each array.  Mean and deviation are
 * accumulated in float64 in a fixed order (the reference: float32 pairwise sums; agreement to float32 rounding) and
 * left as two floats at the start of `workspace` -- fmarl_advantages_workspace() bytes of device memory, 16-byte aligned. */
size_t fmarl_advantages_workspace(void);
int fmarl_advantages(const float *returns, const float *value_preds, const float *active_masks, float *advantages,
                     int64_t count, int denormalize, float mean, float stddev, void *workspace, void *stream);
/* The same in two halves, for data-parallel learners (one rollout shard and one buffer per GPU, gradients averaged -- there is
 * no trajectory to ship): fmarl_advantages_sums writes the raw advantages and leaves (count, sum, sum of squares) of the
 * active entries as three doubles at workspace + 16; the caller adds the triples of all ranks in place (an all-reduce of 24
 * bytes); fmarl_advantages_apply standardises with the mean / deviation of the summed triple -- every rank then normalises
 * with the statistics of the WHOLE batch, as one process holding all envs would (and leaves them at the start of workspace). */
int fmarl_advantages_sums(const float *returns, const float *value_preds, const float *active_masks, float *advantages,
                          int64_t count, int denormalize, float mean, float stddev, void *workspace, void *stream);
int fmarl_advantages_apply(float *advantages, int64_t count, void *workspace, void *stream);

/* The rows of one minibatch of GraphReplayBuffer.feed_forward_generator (mode 0, graph_buffer.py:368-453) or
 * recurrent_generator (mode 1, :597-758) gathered from the buffer: FmarlBatchSrc = the buffer's arrays ((T + 1, n, N, ...)
 * or (T, n, N, ...) as in the reference, adj stored once per env), FmarlBatchDst = the 16 arrays the generator yields, rows
 * first (any NULL is skipped; a non-NULL output needs its source).  share_obs rows are the env's N obs rows back to back
 * and share_agent_id rows 0..N-1 (what the runner's insert stores, graph_mpe_runner.py:470-484) -- neither is kept in the
 * buffer.  env_slot (extra): t * n + env of every row, for a policy that indexes the per-env adj instead of receiving N
 * copies of it.
 *   mode 0: index[r] = flat position of row r over (T, n, N) in C order (the reference's torch.randperm entries).
 *   mode 1: index[j] = a chunk of `chunk` consecutive entries of the (n, N, T)-ordered flat series; rows = chunk * chunks,
 *           row l * chunks + j = entry index[j] * chunk + l; rnn_states / rnn_states_critic have `chunks` rows (the state at
 *           each chunk's first entry). */
typedef struct FmarlBatchSrc {
    const float *obs, *node_obs, *adj_env, *rnn_states, *rnn_states_critic, *actions, *action_log_probs, *value_preds, *returns,
        *masks, *active_masks, *advantages, *available_actions;
    int32_t T, n, N, D, E, F;      /* episode_length, n_rollout_threads, num_agents, obs width, entities, node features */
    int32_t rnn_elems;             /* recurrent_N * hidden_size */
    int32_t act_dim, avail_dim;    /* last axis of actions / action_log_probs, of available_actions */
    int32_t reserved0;
} FmarlBatchSrc;
typedef struct FmarlBatchDst {
    float *share_obs, *obs, *node_obs, *adj;
    int32_t *agent_id, *share_agent_id;
    float *rnn_states, *rnn_states_critic, *actions, *value_preds, *returns, *masks, *active_masks, *old_action_log_probs,
        *adv_targ, *available_actions;
    int64_t *env_slot;
} FmarlBatchDst;
int fmarl_minibatch_gather(const FmarlBatchSrc *src, const FmarlBatchDst *dst, const int64_t *index, int64_t rows, int mode,
                           int chunk, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FMARL_H */
