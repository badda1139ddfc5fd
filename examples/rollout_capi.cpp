// A rollout driven through the C-ABI alone (include/fmarl.h + the HIP runtime): no Python, no torch.
// This is the call sequence a host in any language would bind; it replaces
// the reference's make_train_env + envs.reset() + envs.step(actions) loop
// (reference onpolicy/scripts/train_mpe.py:17-43, onpolicy/runner/shared/graph_mpe_runner.py:35-72).
//
//   rollout_capi [n_envs] [num_agents] [steps] [seed]   -> prints FNV-1a checksums of the final outputs
//
// tests/test_hip_parity.py runs it on the GPU and compares the checksums with the same rollout driven
// through fair_marl_amd.RolloutEngine.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fmarl.h"

#define HIP_OK(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define FMARL_OKAY(call)                                                              \
    do {                                                                              \
        if ((call) != FMARL_OK) { fprintf(stderr, "%s: %s\n", #call, fmarl_last_error()); return 3; } \
    } while (0)

static uint64_t fnv1a(const void *p, size_t bytes) {
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < bytes; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 512, N = argc > 2 ? atoi(argv[2]) : 4;
    const int steps = argc > 3 ? atoi(argv[3]) : 60;
    const uint64_t seed = argc > 4 ? strtoull(argv[4], nullptr, 10) : 3;
    const int O = 3, W = 0, E = 2 * N + O + W, D = 7, F = 11;

    FmarlConfig cfg = {};
    cfg.scenario = FMARL_SCENARIO_NAVIGATION_GRAPH;
    cfg.n_envs = n; cfg.num_agents = N; cfg.num_landmarks = N; cfg.num_obstacles = O; cfg.num_walls = W;
    cfg.episode_length = 25; cfg.has_max_speed = 1; cfg.env_offset = 0; cfg.flags = FMARL_FLAG_ASYNC_RESET;
    cfg.world_size = 2; cfg.max_speed = 2; cfg.collision_rew = 5; cfg.goal_rew = 5; cfg.min_dist_thresh = 0.05;
    cfg.fair_rew = 1; cfg.zeroshift = 5; cfg.max_edge_dist = 1; cfg.min_obs_dist = 0.5; cfg.seed = seed;

    void *h = nullptr;
    FMARL_OKAY(fmarl_create(&cfg, &h));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    void *state = nullptr;
    const size_t state_bytes = fmarl_state_bytes(&cfg);
    HIP_OK(hipMalloc(&state, state_bytes));
    HIP_OK(hipMemsetAsync(state, 0, state_bytes, st));

    const size_t n_obs = (size_t)n * N * D, n_node = (size_t)n * N * E * F, n_adj = (size_t)n * E * E, n_rew = (size_t)n * N;
    FmarlOutputs outs = {};
    HIP_OK(hipMalloc((void **)&outs.obs, n_obs * 4));
    HIP_OK(hipMalloc((void **)&outs.node_obs, n_node * 4));
    HIP_OK(hipMalloc((void **)&outs.adj, n_adj * 4));
    HIP_OK(hipMalloc((void **)&outs.reward, n_rew * 4));
    HIP_OK(hipMalloc((void **)&outs.done, n_rew));
    HIP_OK(hipMalloc((void **)&outs.info, (size_t)FMARL_INFO_WIDTH * n_rew * 4));
    HIP_OK(hipMalloc((void **)&outs.edge_nnz, (size_t)n * 4));   // processAdj fused with the step (gnn.py:307-326)

    // action tape: the high bits of a 64-bit LCG, value % 5 per (step, env, agent)
    std::vector<int32_t> tape((size_t)steps * n * N);
    uint64_t x = 0x9E3779B97F4A7C15ull ^ seed;
    for (auto &a : tape) { x = x * 6364136223846793005ull + 1442695040888963407ull; a = (int32_t)((x >> 33) % 5); }
    int32_t *d_tape = nullptr;
    HIP_OK(hipMalloc((void **)&d_tape, tape.size() * 4));
    HIP_OK(hipMemcpyAsync(d_tape, tape.data(), tape.size() * 4, hipMemcpyHostToDevice, st));

    FMARL_OKAY(fmarl_init_state(h, state, st));                 // make_world x n
    FMARL_OKAY(fmarl_reset(h, state, nullptr, &outs, st));      // envs.reset()
    for (int t = 0; t < steps; ++t)                             // envs.step(actions) incl. the workers' auto-reset
        FMARL_OKAY(fmarl_step(h, state, d_tape + (size_t)t * n * N, nullptr, &outs, 1, st));
    // the policy's edge list of the last step: counts from the emission, prefix sum and COO fill on the device, sized by
    // an upper bound so that nothing has to come back to the host in between
    int64_t *offsets = nullptr, *edge_index = nullptr;
    float *edge_attr = nullptr;
    const int64_t cap = (int64_t)n * E * (E - 1);
    HIP_OK(hipMalloc((void **)&offsets, ((size_t)n + 1) * 8));
    HIP_OK(hipMalloc((void **)&edge_index, (size_t)2 * cap * 8));
    HIP_OK(hipMalloc((void **)&edge_attr, (size_t)cap * 4));
    FMARL_OKAY(fmarl_edge_offsets(outs.edge_nnz, n, 1, offsets, st));
    FMARL_OKAY(fmarl_edge_fill_state(h, state, offsets, edge_index, edge_attr, cap, 1, nullptr, st));
    HIP_OK(hipStreamSynchronize(st));
    int64_t total = 0;
    HIP_OK(hipMemcpy(&total, offsets + n, 8, hipMemcpyDeviceToHost));
    std::vector<int64_t> rows((size_t)total), cols((size_t)total);
    std::vector<float> attr((size_t)total);
    HIP_OK(hipMemcpy(rows.data(), edge_index, (size_t)total * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(cols.data(), edge_index + cap, (size_t)total * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(attr.data(), edge_attr, (size_t)total * 4, hipMemcpyDeviceToHost));
    printf("edges %lld edge_rows %016llx edge_cols %016llx edge_attr %016llx\n", (long long)total,
           (unsigned long long)fnv1a(rows.data(), (size_t)total * 8), (unsigned long long)fnv1a(cols.data(), (size_t)total * 8),
           (unsigned long long)fnv1a(attr.data(), (size_t)total * 4));

    std::vector<float> obs(n_obs), node(n_node), adj(n_adj), rew(n_rew);
    std::vector<uint8_t> done(n_rew);
    HIP_OK(hipMemcpy(obs.data(), outs.obs, n_obs * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(node.data(), outs.node_obs, n_node * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(adj.data(), outs.adj, n_adj * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(rew.data(), outs.reward, n_rew * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(done.data(), outs.done, n_rew, hipMemcpyDeviceToHost));
    printf("obs %016llx node_obs %016llx adj %016llx reward %016llx done %016llx\n",
           (unsigned long long)fnv1a(obs.data(), n_obs * 4), (unsigned long long)fnv1a(node.data(), n_node * 4),
           (unsigned long long)fnv1a(adj.data(), n_adj * 4), (unsigned long long)fnv1a(rew.data(), n_rew * 4),
           (unsigned long long)fnv1a(done.data(), n_rew));

    FMARL_OKAY(fmarl_destroy(h));
    for (void *p : {(void *)outs.obs, (void *)outs.node_obs, (void *)outs.adj, (void *)outs.reward, (void *)outs.done,
                    (void *)outs.info, (void *)outs.edge_nnz, (void *)offsets, (void *)edge_index, (void *)edge_attr, (void *)d_tape, state})
        HIP_OK(hipFree(p));
    HIP_OK(hipStreamDestroy(st));
    return 0;
}
