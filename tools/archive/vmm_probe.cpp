// measurement aid: does it matter for the emission how the output buffers' PHYSICAL memory is put together?
// (profiles/r2_placement_tcc.md: the store stream runs 1.35 ... 1.54 ms per launch depending on the pages the two hipMalloc'd
// buffers got.)  Here node_obs / adj are virtual ranges (hipMemAddressReserve) backed by many small physical chunks
// (hipMemCreate + hipMemMap), mapped in creation order, interleaved between the two buffers, or shuffled -- and the
// emission-only kernel (fmarl_rebuild_graph) is timed on each, next to plain hipMalloc pairs.
//   hipcc -O2 -std=c++17 -Iinclude -o tools/vmm_probe tools/vmm_probe.cpp -Lfair_marl_amd/csrc -lfmarl -Wl,-rpath,$PWD/fair_marl_amd/csrc
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "fmarl.h"
#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); exit(2); } } while (0)
#define F_OK(call) do { if ((call) != FMARL_OK) { fprintf(stderr, "%s: %s\n", #call, fmarl_last_error()); exit(3); } } while (0)

struct Vbuf { void *ptr = nullptr; size_t bytes = 0; std::vector<hipMemGenericAllocationHandle_t> h; };

static size_t gran = 0;
static hipMemAllocationProp prop_() { hipMemAllocationProp p = {}; p.type = hipMemAllocationTypePinned; p.location.type = hipMemLocationTypeDevice; p.location.id = 0; return p; }

// reserve a virtual range and back it with chunks; `order` = physical creation index of virtual chunk k
static void vmap(Vbuf &b, size_t bytes, size_t chunk, std::vector<hipMemGenericAllocationHandle_t> &pool, const std::vector<int> &order) {
    b.bytes = (bytes + chunk - 1) / chunk * chunk;
    HIP_OK(hipMemAddressReserve(&b.ptr, b.bytes, 0, nullptr, 0));
    for (size_t k = 0; k < b.bytes / chunk; ++k) {
        HIP_OK(hipMemMap((char *)b.ptr + k * chunk, chunk, 0, pool[order[k]], 0));
        b.h.push_back(pool[order[k]]);
    }
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    HIP_OK(hipMemSetAccess(b.ptr, b.bytes, &acc, 1));
}
static void vunmap(Vbuf &b) { HIP_OK(hipMemUnmap(b.ptr, b.bytes)); HIP_OK(hipMemAddressFree(b.ptr, b.bytes)); b = Vbuf(); }

int main(int argc, char **argv) {
    const int n = 65536, N = 32, O = 8, E = 2 * N + O, D = 7, F = 11;
    FmarlConfig cfg = {};
    cfg.scenario = FMARL_SCENARIO_NAVIGATION_GRAPH; cfg.n_envs = n; cfg.num_agents = N; cfg.num_landmarks = N; cfg.num_obstacles = O;
    cfg.episode_length = 25; cfg.has_max_speed = 1; cfg.world_size = 2; cfg.max_speed = 2; cfg.collision_rew = 5; cfg.goal_rew = 5;
    cfg.min_dist_thresh = 0.05; cfg.fair_rew = 1; cfg.zeroshift = 5; cfg.max_edge_dist = 1; cfg.min_obs_dist = 0.5; cfg.seed = 1;
    void *h = nullptr; F_OK(fmarl_create(&cfg, &h));
    hipStream_t st; HIP_OK(hipStreamCreate(&st));
    const size_t node_b = (size_t)n * N * E * F * 4, adj_b = (size_t)n * E * E * 4;
    float *obs; void *rec;
    HIP_OK(hipMalloc((void **)&obs, (size_t)n * N * D * 4)); HIP_OK(hipMemset(obs, 0, (size_t)n * N * D * 4));
    const size_t words = fmarl_episode_record_words(&cfg);
    HIP_OK(hipMalloc(&rec, (size_t)n * words * 4)); HIP_OK(hipMemset(rec, 0, (size_t)n * words * 4));
    hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    auto timeit = [&](float *node, float *adj) {
        for (int i = 0; i < 2; ++i) F_OK(fmarl_rebuild_graph(h, obs, rec, n, node, adj, st));
        HIP_OK(hipEventRecord(e0, st));
        for (int i = 0; i < 5; ++i) F_OK(fmarl_rebuild_graph(h, obs, rec, n, node, adj, st));
        HIP_OK(hipEventRecord(e1, st)); HIP_OK(hipEventSynchronize(e1));
        float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
    };
    for (int k = 0; k < 3; ++k) {   // plain hipMalloc pairs
        float *node, *adj; HIP_OK(hipMalloc((void **)&node, node_b)); HIP_OK(hipMalloc((void **)&adj, adj_b));
        printf("hipMalloc pair %d: %.3f ms\n", k, timeit(node, adj)); fflush(stdout);
        if (k < 2) { /* keep them allocated so that the next pair gets other pages */ }
    }
    int vmm = 0; HIP_OK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, 0));
    hipMemAllocationProp prop = prop_();
    HIP_OK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("vmm supported %d, minimum granularity %zu\n", vmm, gran); fflush(stdout);
    if (!vmm) return 0;
    std::mt19937 rng(1);
    for (size_t chunk : {(size_t)256 << 20, (size_t)16 << 20, (size_t)2 << 20}) {
        if (chunk < gran || chunk % gran) continue;
        const size_t kn = (node_b + chunk - 1) / chunk, ka = (adj_b + chunk - 1) / chunk;
        for (int mode = 0; mode < 3; ++mode) {   // 0: creation order, 1: node / adj chunks created alternately, 2: shuffled
            std::vector<hipMemGenericAllocationHandle_t> pool(kn + ka);
            for (auto &hd : pool) HIP_OK(hipMemCreate(&hd, chunk, &prop, 0));
            std::vector<int> on(kn), oa(ka);
            if (mode == 0) { for (size_t k = 0; k < kn; ++k) on[k] = (int)k; for (size_t k = 0; k < ka; ++k) oa[k] = (int)(kn + k); }
            else {
                std::vector<int> all(kn + ka); for (size_t k = 0; k < kn + ka; ++k) all[k] = (int)k;
                if (mode == 2) std::shuffle(all.begin(), all.end(), rng);
                // mode 1: adj chunks spread evenly between the node chunks (every (kn + ka) / ka -th physical chunk)
                size_t ia = 0, in = 0;
                for (size_t k = 0; k < kn + ka; ++k) {
                    const bool to_adj = ia < ka && (in >= kn || (k * ka) / (kn + ka) >= ia);
                    if (to_adj) oa[ia++] = all[k]; else on[in++] = all[k];
                }
            }
            Vbuf node, adj; vmap(node, node_b, chunk, pool, on); vmap(adj, adj_b, chunk, pool, oa);
            const float ms = timeit((float *)node.ptr, (float *)adj.ptr);
            printf("vmm chunk %4zu MiB mode %d (%s): %.3f ms\n", chunk >> 20, mode, mode == 0 ? "creation order" : (mode == 1 ? "interleaved" : "shuffled"), ms); fflush(stdout);
            vunmap(node); vunmap(adj);
            for (auto &hd : pool) HIP_OK(hipMemRelease(hd));
        }
    }
    return 0;
}
