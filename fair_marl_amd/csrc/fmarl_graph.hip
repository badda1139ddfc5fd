// Scenario.update_graph on gfx950 (reference multiagent/custom_scenarios/navigation_graph.py:1037-1056):
// connect = (dist <= max_edge_dist) & (dist > 0) -> COO edge list in row-major order (scipy csr -> coo)
// + edge weights.  One wave per environment: 64 entries per pass, ballot + popcount stream compaction.
#pragma once
#include "fmarl_dev.h"
#include "fmarl_kernels.h"

namespace fmarl {

__global__ __launch_bounds__(256) void update_graph_kernel(const float *adj, int32_t *edge_index, float *edge_weight,
                                                           int32_t *nnz, int n_envs, int E, float max_edge_dist) {
    const int lane = threadIdx.x & 63;
    const int env = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (env >= n_envs) return;   // wave-uniform
    const int EE = E * E;
    const float *a = adj + (size_t)env * EE;
    int32_t *rows = edge_index + (size_t)env * 2 * EE, *cols = rows + EE;
    float *wts = edge_weight + (size_t)env * EE;
    int base = 0;
    for (int q0 = 0; q0 < EE; q0 += 64) {
        const int q = q0 + lane;
        const float d = q < EE ? a[q] : 0.f;
        const bool on = q < EE && d <= max_edge_dist && d > 0.f;
        const unsigned long long m = __ballot(on);
        if (on) {
            const int k = base + __popcll(m & ((1ull << lane) - 1));
            rows[k] = q / E; cols[k] = q % E; wts[k] = d;
        }
        base += __popcll(m);
    }
    for (int k = base + lane; k < EE; k += 64) { rows[k] = -1; cols[k] = -1; wts[k] = 0.f; }
    if (lane == 0) nnz[env] = base;
}

}  // namespace fmarl
