// Kernel entry points of libfmarl (gfx950).  Launch geometry is chosen in fmarl_capi.hip.
#pragma once
#include "fmarl_dev.h"

namespace fmarl {

// fmarl_step.hip
template <int SH> __global__ void step_kernel(Params p, FmarlOutputs o, const int32_t *action_idx, const float *action_vec,
                                              int auto_reset);
template <int SH> __global__ void step_end_kernel(Params p, FmarlOutputs o, const int32_t *action_idx, const float *action_vec,
                                                  int auto_reset);
__device__ void emit_graph(const Params &p, const FmarlOutputs &o, char *lds, int env0, int nenv);
__device__ void load_statics(const Params &p, char *lds, int env0, int nenv);
__device__ void load_statics_range(const Params &p, char *lds, int env0, int el_begin, int el_end, int thr, int nthr);

// fmarl_reset.hip
enum ResetMode { kResetAll = 0, kResetMask = 1, kResetAuto = 2, kResetInit = 3, kResetStage = 4 };
__global__ void reset_commit_kernel(Params p, int mode, const uint8_t *mask);
__global__ void stage_finish_kernel(Params p);
template <bool LDS> __global__ void reset_place_kernel(Params p, int mode, const uint8_t *mask);
__global__ void reset_emit_kernel(Params p, FmarlOutputs o);
__global__ void cost_matrix_kernel(const double2 *agent_pos, const double2 *goal_pos, double *costs,
                                   int n_envs, int N, int L);

// fmarl_formation.hip
template <bool STEP, int SH> __global__ void formation_kernel(Params p, FmarlOutputs o, const int32_t *action_idx,
                                                     const float *action_vec, int auto_reset);

__global__ void formation_rebuild_kernel(Params p, FmarlOutputs o, const uint32_t *ep_rec, const uint32_t *step_rec, int n_envs);

// fmarl_fairnav.hip
template <bool STEP, int THREADS, int NL> __global__ void fairnav_kernel(Params p, FmarlOutputs o, const int32_t *action_idx,
                                                                const float *action_vec, int auto_reset);

template <int THREADS> __global__ void fairnav_rebuild_kernel(Params p, FmarlOutputs o, const uint32_t *ep_rec, const uint32_t *step_rec, int n_envs);

// fmarl_lexifair.hip
void launch_lexifair_costs(const double *costs, int32_t *perm, int n_envs, int N, hipStream_t stream);
void launch_lexifair_state(const Params &p, hipStream_t stream);

// fmarl_graph.hip
__global__ void update_graph_kernel(const float *adj, int32_t *edge_index, float *edge_weight, int32_t *nnz,
                                    int n_envs, int E, float max_edge_dist);

__global__ void update_graph_state_kernel(Params p, int32_t *edge_index, double *edge_weight, int32_t *nnz, double max_edge_dist);

__global__ void edge_scan_totals_kernel(const int32_t *nnz, int n_graphs, int gpe, int64_t *offsets);
__global__ void edge_scan_chunks_kernel(int n_graphs, int64_t *offsets);
__global__ void edge_scan_fill_kernel(const int32_t *nnz, int n_graphs, int gpe, int64_t *offsets);
__global__ void edge_fill_state_kernel(Params p, const int64_t *offsets, int64_t *edge_index, float *edge_attr, int64_t capacity, int gpe);
__global__ void info_mean_kernel(const float *info, double *out, int n_envs, int N, double unreached_time);
__global__ void edge_count_kernel(const float *adj, int32_t *nnz, int n_envs, int E, float thr, int strict);
__global__ void edge_fill_kernel(const float *adj, const int64_t *offsets, int64_t *edge_index, float *edge_attr,
                                 int64_t total, int n_graphs, int graphs_per_env, int E, float thr, int strict);

}  // namespace fmarl
