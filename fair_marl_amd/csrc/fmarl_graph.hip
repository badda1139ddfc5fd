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

// Position of entity e of one env in the reference's entity order (multiagent/core.py:179-186: agents, landmarks,
// obstacles, walls; a wall's p_pos is (0, axis) for 'H' and (axis, 0) for 'V', navigation_graph.py:309-324).
__device__ __forceinline__ double2 entity_pos(const Params &p, int env, int e) {
    if (e < p.N) return p.agent_pos[(size_t)env * p.N + e];
    e -= p.N;
    if (e < p.L) return p.landmark_pos[(size_t)env * p.L + e];
    e -= p.L;
    if (e < p.O) return p.obstacle_pos[(size_t)env * p.O + e];
    e -= p.O;
    const double axis = p.wall_axis[(size_t)env * p.W + e];
    return p.wall_orient[(size_t)env * p.W + e] == 0 ? make_double2(0.0, axis) : make_double2(axis, 0.0);
}

// Scenario.update_graph exactly as the reference computes it (navigation_graph.py:1037-1056 on the float64
// cached_dist_mag of multiagent/core.py:204-228, np.linalg.norm(axis=2) = sqrt(dx*dx + dy*dy)): distances from the f64
// state, `<=` against the f64 threshold, f64 weights -- the f32 adj output cannot decide a distance within a float32
// ulp of max_edge_dist.  One wave per env, row-major ballot + popcount compaction.  Not on the rollout's hot path
// (the reference's only consumer is the renderer, environment.py:491-506).
__global__ __launch_bounds__(256) void update_graph_state_kernel(Params p, int32_t *edge_index, double *edge_weight,
                                                                 int32_t *nnz, double max_edge_dist) {
    const int lane = threadIdx.x & 63;
    const int env = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (env >= p.n_envs) return;   // wave-uniform
    const int E = p.E, EE = E * E;
    int32_t *rows = edge_index + (size_t)env * 2 * EE, *cols = rows + EE;
    double *wts = edge_weight + (size_t)env * EE;
    int base = 0;
    for (int q0 = 0; q0 < EE; q0 += 64) {
        const int q = q0 + lane;
        double d = 0.0;
        int r = 0, c = 0;
        if (q < EE) {
            r = q / E; c = q - r * E;
            const double2 a = entity_pos(p, env, r), b = entity_pos(p, env, c);
            const double dx = a.x - b.x, dy = a.y - b.y;
            d = sqrt(dx * dx + dy * dy);
        }
        const bool on = q < EE && d <= max_edge_dist && d > 0.0;
        const unsigned long long m = __ballot(on);
        if (on) {
            const int k = base + __popcll(m & ((1ull << lane) - 1));
            rows[k] = r; cols[k] = c; wts[k] = d;
        }
        base += __popcll(m);
    }
    for (int k = base + lane; k < EE; k += 64) { rows[k] = -1; cols[k] = -1; wts[k] = 0.0; }
    if (lane == 0) nnz[env] = base;
}

// Policy-side edge construction, reference onpolicy/algorithms/utils/gnn.py:307-326 processAdj + the
// PyG batching of :243-253: per graph b (graph b uses the adj of env b / graphs_per_env -- the reference
// feeds every agent its own copy of the same matrix), edges (r, c) with 0 < adj[r][c] < max_edge_dist
// (strict, unlike update_graph) in row-major order, node ids offset by b * E.  Two passes around a prefix
// sum of the per-graph counts: edge_count_kernel -> offsets (caller: cumsum) -> edge_fill_kernel.
__device__ __forceinline__ bool edge_on(float d, float thr, int strict) { return d > 0.f && (strict ? d < thr : d <= thr); }

__global__ __launch_bounds__(256) void edge_count_kernel(const float *adj, int32_t *nnz, int n_envs, int E, float thr, int strict) {
    const int lane = threadIdx.x & 63;
    const int env = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (env >= n_envs) return;   // wave-uniform
    const float *a = adj + (size_t)env * E * E;
    int cnt = 0;
    for (int q = lane; q < E * E; q += 64) cnt += edge_on(a[q], thr, strict) ? 1 : 0;
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) nnz[env] = cnt;
}

__global__ __launch_bounds__(256) void edge_fill_kernel(const float *adj, const int64_t *offsets, int64_t *edge_index,
                                                        float *edge_attr, int64_t total, int n_graphs, int graphs_per_env,
                                                        int E, float thr, int strict) {
    const int lane = threadIdx.x & 63;
    const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // one wave per graph
    if (b >= n_graphs) return;
    const int EE = E * E;
    const float *a = adj + (size_t)(b / graphs_per_env) * EE;
    int64_t *rows = edge_index, *cols = edge_index + total;
    int64_t base = offsets[b];
    const int64_t end = offsets[b + 1], limit = end < total ? end : total;   // (a graph stays inside its own range)
    const int64_t node0 = (int64_t)b * E;
    for (int q0 = 0; q0 < EE; q0 += 64) {
        const int q = q0 + lane;
        const float d = q < EE ? a[q] : 0.f;
        const bool on = q < EE && edge_on(d, thr, strict);
        const unsigned long long m = __ballot(on);
        if (on) {
            const int64_t k = base + __popcll(m & ((1ull << lane) - 1));
            if (k < limit) { rows[k] = node0 + q / E; cols[k] = node0 + q % E; edge_attr[k] = d; }   // total = capacity of the buffers
        }
        base += __popcll(m);
    }
}

// Exclusive prefix sum of the per-graph edge counts, int32 counts -> int64 offsets (n_graphs + 1), graph b counting
// nnz[b / graphs_per_env].  Three small launches, no scratch memory: the chunk totals are parked in the offsets array
// itself, at the position where the prefix of the NEXT chunk belongs.
constexpr int kScanChunk = 2048;   // graphs per workgroup (256 threads x 8)

__global__ __launch_bounds__(256) void edge_scan_totals_kernel(const int32_t *nnz, int n_graphs, int gpe, int64_t *offsets) {
    __shared__ int64_t part[256];
    const int c0 = blockIdx.x * kScanChunk;
    int64_t s = 0;
    for (int k = threadIdx.x; k < kScanChunk; k += 256) {
        const int b = c0 + k;
        if (b < n_graphs) s += nnz[b / gpe];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[min(c0 + kScanChunk, n_graphs)] = part[0];
}

// one workgroup: running sum over the chunk boundaries (at most a few thousand), offsets[0] = 0
__global__ __launch_bounds__(256) void edge_scan_chunks_kernel(int n_graphs, int64_t *offsets) {
    __shared__ int64_t part[256];
    __shared__ int64_t carry;
    const int n_chunks = (n_graphs + kScanChunk - 1) / kScanChunk;
    if (threadIdx.x == 0) { carry = 0; offsets[0] = 0; }
    __syncthreads();
    for (int base = 0; base < n_chunks; base += 256) {
        const int c = base + threadIdx.x;
        const size_t pos = (size_t)min((c + 1) * kScanChunk, n_graphs);
        const int64_t v = c < n_chunks ? offsets[pos] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {   // inclusive Hillis-Steele scan of the 256 totals
            const int64_t add = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (c < n_chunks) offsets[pos] = carry + part[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void edge_scan_fill_kernel(const int32_t *nnz, int n_graphs, int gpe, int64_t *offsets) {
    __shared__ int64_t part[256];
    const int c0 = blockIdx.x * kScanChunk, t0 = c0 + threadIdx.x * 8;
    int64_t v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int b = t0 + k; v[k] = b < n_graphs ? nnz[b / gpe] : 0; s += v[k]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int64_t add = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    int64_t run = offsets[c0] + part[threadIdx.x] - s;   // prefix of this thread's first graph
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int b = t0 + k;
        // (the boundary entries c0 and c0 + kScanChunk already hold their final values)
        if (b < n_graphs && b != c0) offsets[b] = run;
        run += v[k];
    }
}

// processAdj from the world state: one wave per graph recomputes the env's adj entries exactly as
// the emission did -- differences of the float32 roundings of the entity positions -- and compacts the policy edges 0 < d < max_edge_dist in row-major
// order behind offsets[b]: rows | cols with node ids b * E + r, edge_attr = the adj entry.  adj is not read.
// A graph never writes outside its own range [offsets[b], offsets[b + 1]): if the counts the offsets were built from do not
// describe this state (another output set's edge_nnz, a state written in between), the graph's list is cut short or keeps
// a gap instead of running into its neighbour's, and `mismatch` (optional device counter) counts such graphs.
__global__ __launch_bounds__(256) void edge_fill_state_kernel(Params p, const int64_t *offsets, int64_t *edge_index,
                                                              float *edge_attr, int64_t capacity, int gpe, int32_t *mismatch) {
    const int lane = threadIdx.x & 63;
    const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= p.n_envs * gpe) return;   // wave-uniform
    const int env = b / gpe, E = p.E, EE = E * E;
    int64_t *rows = edge_index, *cols = edge_index + capacity;
    int64_t base = offsets[b];
    const int64_t end = offsets[b + 1], limit = end < capacity ? end : capacity;
    const int64_t node0 = (int64_t)b * E;
    for (int q0 = 0; q0 < EE; q0 += 64) {
        const int q = q0 + lane;
        float d = 0.f;
        int r = 0, c = 0;
        if (q < EE) {
            r = q / E; c = q - r * E;
            const double2 a = entity_pos(p, env, r), bb = entity_pos(p, env, c);
            d = dist_f32((float)a.x - (float)bb.x, (float)a.y - (float)bb.y);
        }
        const bool on = q < EE && d > 0.f && d < p.edge_thr;
        const unsigned long long m = __ballot(on);
        if (on) {
            const int64_t k = base + __popcll(m & ((1ull << lane) - 1));
            if (k < limit) { rows[k] = node0 + r; cols[k] = node0 + c; edge_attr[k] = d; }
        }
        base += __popcll(m);
    }
    if (mismatch && lane == 0 && base != end) atomicAdd(mismatch, 1);
}

// Per-agent means over the envs of every info field, reference onpolicy/runner/shared/base_runner.py:197-276
// process_infos + :291-306 log_env (np.mean of each per-agent list); Time_req_to_goal == -1 counts as
// episode_length * dt (:212-215).  One workgroup per (field, agent); f64 tree reduction, deterministic.
__global__ __launch_bounds__(256) void info_mean_kernel(const float *info, double *out, int n_envs, int N, double unreached_time) {
    __shared__ double part[256];
    const int k = blockIdx.x / N, a = blockIdx.x - k * N;
    const float *src = info + (size_t)k * n_envs * N + a;
    double s = 0.0;
    for (int e = threadIdx.x; e < n_envs; e += 256) {
        double v = (double)src[(size_t)e * N];
        if (k == FMARL_INFO_TIME_REQ_TO_GOAL && v == -1.0) v = unreached_time;
        s += v;
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = part[0] / n_envs;
}

// Measurement aid (fmarl_store_stream): pure 16-byte store streams over `n16` 16-byte words -- what the box's HBM takes when a kernel
// does nothing but write.  The step kernels' output stream cannot beat the best of these on the same byte count (bench.py
// `store_ceiling_ms`).  Shapes: 0 = flat grid-stride stream (consecutive workgroups write consecutive 4 KiB, 2 048 workgroups);
// 1 = a workgroup streams a contiguous chunk of `chunk16` words front to back, 4 KiB per instruction round (the shape of the
// adj emission); 2 = every WAVE streams its own contiguous quarter of the workgroup's chunk, 1 KiB per store instruction (the
// shape of the node_obs emission: a wave owns an env's rows).  Workgroup b takes chunk (b * order) mod n_chunks, then -- `persist`
// workgroups that live for the whole launch, as a span's do -- chunk ((b + k grid) * order) mod n_chunks for k = 1, 2, ...: order 1
// is dispatch order (the resident workgroups write one compact window), a large odd order scatters them over the buffer.
// The value depends on the address, so no two stores are equal.
// masks / active_masks of the runner's insert (onpolicy/runner/shared/graph_mpe_runner.py:444-465) for `rows` env-steps of N agents:
// masks = 0 where the agent is done; active_masks = 0 where the agent is done but its env is not (an env whose agents are all
// done keeps active_masks 1).  One lane per env-step: N bytes in, 2 N floats out (a rollout buffer's per-step bookkeeping as one
// launch instead of eight elementwise ones -- at 3 agents x 4 096 envs those cost three times the step kernel).
__global__ __launch_bounds__(256) void insert_masks_kernel(const uint8_t *done, float *masks, float *active, size_t rows, int N) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const uint8_t *d = done + r * N;
    bool all = true;
    for (int i = 0; i < N; ++i) all &= d[i] != 0;
    for (int i = 0; i < N; ++i) {
        const bool di = d[i] != 0;
        masks[r * N + i] = di ? 0.f : 1.f;
        active[r * N + i] = (di && !all) ? 0.f : 1.f;
    }
}

typedef float fmarl_f4 __attribute__((ext_vector_type(4)));
template <int SHAPE>   // shapes 3 / 4: 1 / 2 with non-temporal stores (experiments: no faster than plain stores on MI355X)
__global__ __launch_bounds__(256) void store_stream_kernel(float4 *dst, size_t n16, uint32_t chunk16, uint32_t n_chunks, uint32_t order) {
    auto put = [&](size_t i) {
        if (SHAPE >= 3) { const fmarl_f4 w = {(float)(uint32_t)i, 1.f, 2.f, 3.f}; __builtin_nontemporal_store(w, (fmarl_f4 *)&dst[i]); }
        else dst[i] = make_float4((float)(uint32_t)i, 1.f, 2.f, 3.f);
    };
    if (SHAPE == 0) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) put(i);
        return;
    }
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint32_t chunk = (uint32_t)(((uint64_t)c * order) % n_chunks);
        const size_t b0 = (size_t)chunk * chunk16, be = b0 + chunk16 < n16 ? b0 + chunk16 : n16;
        if (SHAPE == 1 || SHAPE == 3) {
            for (size_t i = b0 + threadIdx.x; i < be; i += 256) put(i);
        } else {
            const uint32_t per_wave = (chunk16 + 3) / 4, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            const size_t b = b0 + (size_t)wave * per_wave, e = b + per_wave < be ? b + per_wave : be;
            for (size_t i = b + lane; i < e; i += 64) put(i);
        }
    }
}

// fmarl_store_pattern: the store stream of the GENERIC emission path (row shapes whose width is not a multiple of 16 bytes: 10 agents,
// E = 23 -> 1 012-byte ego rows) with everything but the stores removed.  A workgroup owns group (b * order) mod groups of every time
// slot: `node_group` contiguous bytes of node rows, which its four waves write window by window -- `window` bytes (64 rows) at 4-byte
// aligned starts, as aligned 16-byte chunks per lane plus up to three dwords at either end, flush_rows' shape -- and `adj_group`
// contiguous bytes of adjacency words, each wave a contiguous quarter, one dword per lane and store (emit_adj_generic's shape); odd
// workgroups write the adjacency first (emit_graph).  The workgroup walks the slots in order, as a span kernel walks its steps.
__global__ __launch_bounds__(256) void store_pattern_kernel(char *node, char *adj, size_t node_group, size_t adj_group, uint32_t groups, uint32_t slots,
                                                            size_t node_slot, size_t adj_slot, uint32_t window, uint32_t order) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t g = (uint32_t)(((uint64_t)blockIdx.x * order) % groups);
    const bool adj_first = (blockIdx.x & 1) != 0;
    for (uint32_t t = 0; t < slots; ++t) {
        for (int pass = 0; pass < 2; ++pass) {
            if ((pass == 0) == adj_first) {   // adjacency words
                const size_t off = (size_t)g * adj_group, size = off < adj_slot ? (adj_slot - off < adj_group ? adj_slot - off : adj_group) : 0;
                const uint32_t words = (uint32_t)(size >> 2), per = (words + 3) / 4, w0 = wave * per < words ? wave * per : words, w1 = w0 + per < words ? w0 + per : words;
                float *d = (float *)(adj + (size_t)t * adj_slot + off);
                for (uint32_t k = w0 + lane; k < w1; k += 64) d[k] = (float)k;
            } else {                          // node rows, a window per wave and round
                const size_t off = (size_t)g * node_group, size = off < node_slot ? (node_slot - off < node_group ? node_slot - off : node_group) : 0;
                char *base = node + (size_t)t * node_slot + off;
                for (size_t w = (size_t)wave * window; w < size; w += (size_t)4 * window) {
                    char *start = base + w, *end = base + (w + window < size ? w + window : size);
                    char *a0 = (char *)(((uintptr_t)start + 15) & ~(uintptr_t)15), *a1 = (char *)((uintptr_t)end & ~(uintptr_t)15);
                    for (char *q = a0 + lane * 16; q < a1; q += 64 * 16) *(float4 *)q = make_float4(1.f, 2.f, 3.f, (float)lane);
                    if (lane < 3) { char *q = start + lane * 4; if (q < (a0 < end ? a0 : end)) *(float *)q = 4.f; }
                    else if (lane < 6) { char *q = (a1 > a0 ? a1 : a0) + (lane - 3) * 4; if (q < end) *(float *)q = 5.f; }
                }
            }
        }
    }
}

// fmarl_ring_alloc's check of an array allocated after another one was freed: a KERNEL's fill read back by a kernel (the faults seen on
// re-used address ranges passed a fill / read-back through the copy engines and lost a kernel's writes).  Word i gets a pattern of i;
// the second launch counts the words that do not hold it and zeroes the array.
__global__ __launch_bounds__(256) void ring_fill_kernel(uint4 *dst, size_t n16, uint32_t salt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint32_t w = (uint32_t)i * 2654435761u ^ salt;
        dst[i] = make_uint4(w, ~w, w + 1u, salt);
    }
}
__global__ __launch_bounds__(256) void ring_check_kernel(uint4 *dst, size_t n16, uint32_t salt, unsigned long long *bad) {
    unsigned long long mine = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint32_t w = (uint32_t)i * 2654435761u ^ salt;
        const uint4 v = dst[i];
        mine += (v.x != w) + (v.y != ~w) + (v.z != w + 1u) + (v.w != salt);
        dst[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (mine) atomicAdd(bad, mine);
}

// test hook (fmarl_poison_lds): every workgroup writes 0xFF bytes over all the LDS it was given
__global__ __launch_bounds__(256) void poison_lds_kernel(int words) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int k = threadIdx.x; k < words; k += 256) ((volatile uint32_t *)lds)[k] = 0xFFFFFFFFu;
}

}  // namespace fmarl
