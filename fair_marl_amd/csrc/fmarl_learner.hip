// The learner's side of the rollout buffer on gfx950: what the reference's trainer computes on the
// filled GraphReplayBuffer with NumPy on the host before the first gradient step --
//   * GraphReplayBuffer.compute_returns         (onpolicy/utils/graph_buffer.py:285-366)   -> returns_kernel
//   * the standardised advantages of GR_MAPPO.train (onpolicy/algorithms/graph_mappo.py:294-304) -> advantage_*_kernel
//   * feed_forward_generator / recurrent_generator (graph_buffer.py:368-453, 597-758)     -> minibatch_gather_kernel
// All of it is float32 streaming over (T, n, N) arrays: HBM-bound, no reuse, nothing for LDS beyond a block reduction.
#pragma once
#include "fmarl_dev.h"
#include "fmarl_kernels.h"

namespace fmarl {

// One thread per (env, agent) column walks the time axis backwards; consecutive threads read consecutive columns, so
// every load / store of a step is one contiguous segment per wave.  Float32 in the reference's order of operations
// (-ffp-contract=off: no fused multiply-add): the result equals NumPy's bit for bit in all twelve branches
// (tests/golden/learner_*.npz).  `denorm`: value_normalizer.denormalize = x * stddev + mean, product rounded first
// (onpolicy/utils/valuenorm.py:92-104, onpolicy/algorithms/utils/popart.py:101-111).
__global__ __launch_bounds__(256) void returns_kernel(FmarlReturns a, const float *rewards, float *value_preds, const float *masks,
                                                      const float *bad_masks, const float *next_value, float *returns) {
    const size_t C = (size_t)a.columns;
    const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int T = a.T;
    const float g = (float)a.gamma, gl = (float)(a.gamma * a.gae_lambda);   // gamma * gae_lambda is formed in double first
    const float mean = a.mean, sd = a.stddev;
    const bool dn = a.denormalize != 0, proper = a.use_proper_time_limits != 0;
    const float nv = next_value[c];
    if (a.use_gae) {
        value_preds[(size_t)T * C + c] = nv;                 // :299 / :340
        float v1 = dn ? nv * sd + mean : nv;                 // denormalised value of step t + 1
        float gae = 0.f;
        float r = T > 0 ? rewards[(size_t)(T - 1) * C + c] : 0.f, v = T > 0 ? value_preds[(size_t)(T - 1) * C + c] : 0.f;
        float m1 = masks[(size_t)T * C + c], b1 = proper ? bad_masks[(size_t)T * C + c] : 1.f;
        for (int t = T - 1; t >= 0; --t) {
            // the next iteration's operands are requested before this one's arithmetic
            float rn = 0.f, vn = 0.f, mn = 0.f, bn = 1.f;
            if (t > 0) {
                rn = rewards[(size_t)(t - 1) * C + c];
                vn = value_preds[(size_t)(t - 1) * C + c];
                mn = masks[(size_t)t * C + c];
                if (proper) bn = bad_masks[(size_t)t * C + c];
            }
            const float v0 = dn ? v * sd + mean : v;
            const float delta = (r + (g * v1) * m1) - v0;                   // :305-309 / :316-318 / :344-347 / :353-355
            gae = (proper && dn) ? delta + (gl * gae) * m1                  // :310-311
                                 : delta + (gl * m1) * gae;                 // :319-320 / :348-349 / :356-357
            if (proper) gae = gae * b1;                                     // :312 / :321
            returns[(size_t)t * C + c] = gae + v0;                          // :313-314 / :322 / :350-351 / :358
            v1 = v0; r = rn; v = vn; m1 = mn; b1 = bn;
        }
    } else {
        float acc = nv;
        returns[(size_t)T * C + c] = nv;                     // :324 / :360
        for (int t = T - 1; t >= 0; --t) {
            acc = (acc * g) * masks[(size_t)(t + 1) * C + c] + rewards[(size_t)t * C + c];   // :327-328 / :333-334 / :362-364
            if (proper) {
                const float b = bad_masks[(size_t)(t + 1) * C + c], v = value_preds[(size_t)t * C + c];
                acc = acc * b + (1.f - b) * (dn ? v * sd + mean : v);       // :329-331 / :335-337
            }
            returns[(size_t)t * C + c] = acc;
        }
    }
}

// advantages = returns[:-1] - denormalised value_preds[:-1]; mean / std over the entries whose active mask is not 0
// (np.nanmean / np.nanstd after the masked entries were set to NaN, graph_mappo.py:300-303).  Pass 1 writes the raw
// advantages and per-workgroup (count, sum, sum of squares) in float64.
// (count, sum, sum of squares) of a workgroup of four waves, the same value in every thread; the order of the additions is fixed
__device__ __forceinline__ void block_sum3(double &a, double &b, double &c, double (*red)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 32; off; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); c += __shfl_down(c, off); }
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = c; }
    __syncthreads();
    a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    c = red[2][0] + red[2][1] + red[2][2] + red[2][3];
}

template <int V>   // V = 4: 16-byte lanes (all four arrays 16-byte aligned), V = 1: any alignment
__global__ __launch_bounds__(256) void advantage_raw_kernel(const float *returns, const float *value_preds, const float *active_masks,
                                                            float *adv, size_t total, float mean, float sd, int dn, double *partials) {
    __shared__ double red[3][4];
    double cnt = 0.0, s1 = 0.0, s2 = 0.0;
    auto one = [&](float r, float v, float m) {
        const float x = r - (dn ? v * sd + mean : v);
        if (m != 0.f && x == x) { cnt += 1.0; s1 += (double)x; s2 += (double)x * (double)x; }   // (np.nanmean / np.nanstd skip a NaN advantage too)
        return x;
    };
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
    if (V == 4) {
        // two 16-byte chunks per array in flight per lane before the first store: on gfx9 a load issued behind a store waits
        // for that store's acknowledgement (one counter orders both), so a load - store - load chain pays both round trips
        const float4 *r4 = (const float4 *)returns, *v4 = (const float4 *)value_preds, *m4 = (const float4 *)active_masks;
        float4 *a4 = (float4 *)adv;
        const size_t n4 = total >> 2;
        size_t i = tid;
        for (; i + nthreads < n4; i += 2 * nthreads) {
            const float4 ra = r4[i], va = v4[i], ma = m4[i], rb = r4[i + nthreads], vb = v4[i + nthreads], mb = m4[i + nthreads];
            a4[i] = make_float4(one(ra.x, va.x, ma.x), one(ra.y, va.y, ma.y), one(ra.z, va.z, ma.z), one(ra.w, va.w, ma.w));
            a4[i + nthreads] = make_float4(one(rb.x, vb.x, mb.x), one(rb.y, vb.y, mb.y), one(rb.z, vb.z, mb.z), one(rb.w, vb.w, mb.w));
        }
        if (i < n4) {
            const float4 ra = r4[i], va = v4[i], ma = m4[i];
            a4[i] = make_float4(one(ra.x, va.x, ma.x), one(ra.y, va.y, ma.y), one(ra.z, va.z, ma.z), one(ra.w, va.w, ma.w));
        }
        const size_t k = (n4 << 2) + tid;   // the last total % 4 entries
        if (k < total) adv[k] = one(returns[k], value_preds[k], active_masks[k]);
    } else {
        for (size_t i = tid; i < total; i += nthreads) adv[i] = one(returns[i], value_preds[i], active_masks[i]);
    }
    block_sum3(cnt, s1, s2, red);
    if (threadIdx.x == 0) {
        double *mine = partials + 3 * (size_t)blockIdx.x;
        mine[0] = cnt; mine[1] = s1; mine[2] = s2;
    }
}

// The partials of pass 1 as ONE triple (count, sum, sum of squares): for a caller that adds the triples of several processes
// (data-parallel learners, one rollout shard per GPU) before pass 2 -- fmarl_advantages_sums / fmarl_advantages_apply.
__global__ __launch_bounds__(256) void advantage_sum_kernel(const double *partials, int nparts, double *sums) {
    __shared__ double red[3][4];
    double cnt = 0.0, s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nparts; b += blockDim.x) {
        const double *q = partials + 3 * (size_t)b;
        cnt += q[0]; s1 += q[1]; s2 += q[2];
    }
    block_sum3(cnt, s1, s2, red);
    if (threadIdx.x == 0) { sums[0] = cnt; sums[1] = s1; sums[2] = s2; }
}

// Pass 2: every workgroup adds the partials of pass 1 in the same fixed order (thread k takes k, k + 256, ...; then the wave
// and block reduction) -- a few KB out of L2, and no fence or atomic ticket: an agent-scope release on this chip writes the
// XCD's L2 back, which cost more than the whole pass -- and standardises its share of the entries.  Workgroup 0 leaves
// (mean, std) in stats[0..1].
__global__ __launch_bounds__(256) void advantage_scale_kernel(float *adv, size_t total, const double *partials, int nparts, float *stats) {
    __shared__ double red[3][4];
    double cnt = 0.0, s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nparts; b += blockDim.x) {
        const double *q = partials + 3 * (size_t)b;
        cnt += q[0]; s1 += q[1]; s2 += q[2];
    }
    block_sum3(cnt, s1, s2, red);
    // no active entry at all: np.nanmean / np.nanstd of an all-NaN array give NaN (graph_mappo.py:302-304), and so does this
    const double mu = cnt > 0.0 ? s1 / cnt : __longlong_as_double(0x7ff8000000000000ll);
    const double var = cnt > 0.0 ? s2 / cnt - mu * mu : mu;
    const float m = (float)mu, sdev = (float)(var != var ? var : sqrt(var > 0.0 ? var : 0.0));
    if (blockIdx.x == 0 && threadIdx.x == 0) { stats[0] = m; stats[1] = sdev; }
    const float d = sdev + 1e-5f;   // graph_mappo.py:304
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)adv) & 15) == 0) {
        float4 *a4 = (float4 *)adv;
        const size_t n4 = total >> 2;
        size_t i = tid;
        for (; i + nthreads < n4; i += 2 * nthreads) {
            float4 a = a4[i], b = a4[i + nthreads];
            a.x = (a.x - m) / d; a.y = (a.y - m) / d; a.z = (a.z - m) / d; a.w = (a.w - m) / d;
            b.x = (b.x - m) / d; b.y = (b.y - m) / d; b.z = (b.z - m) / d; b.w = (b.w - m) / d;
            a4[i] = a; a4[i + nthreads] = b;
        }
        if (i < n4) {
            float4 a = a4[i];
            a.x = (a.x - m) / d; a.y = (a.y - m) / d; a.z = (a.z - m) / d; a.w = (a.w - m) / d;
            a4[i] = a;
        }
        const size_t k = (n4 << 2) + tid;
        if (k < total) adv[k] = (adv[k] - m) / d;
    } else {
        for (size_t i = tid; i < total; i += nthreads) adv[i] = (adv[i] - m) / d;
    }
}

// Rows of a minibatch: one wave per tile of (up to) 64 output rows.  Lane l first resolves row r0 + l to its buffer cell, so the
// per-row scalars (value_preds, returns, masks, ...) leave as one coalesced store per field and the narrow fields (obs,
// actions, available_actions, ...: a few floats per row) as one contiguous run for the whole tile; the wide fields
// (share_obs, node_obs, adj, rnn states) are copied row by row with all 64 lanes on one row, 16 bytes per lane where the row
// width allows.  mode 0 (feed_forward_generator): index[r] is the flat position of row r over (T, n, N) in C order.
// mode 1 (recurrent_generator): index[j] is a chunk of `chunk` consecutive entries of the (n, N, T)-ordered series; output
// row l * chunks + j is entry index[j] * chunk + l (a chunk runs over the end of one (env, agent) series into the next,
// exactly as the reference's reshape does), and the two rnn-state outputs have one row per chunk: the state at the chunk's
// first entry.
// four loads in flight per lane before the first store (a wave copies one row at a time: without this a row's worth of
// bandwidth is one 256-byte request per round trip)
template <typename V>
__device__ __forceinline__ void copy_vec(V *__restrict__ dst, const V *__restrict__ src, int count, int lane) {
    int k = lane;
    for (; k + 192 < count; k += 256) {
        const V a = src[k], b = src[k + 64], c = src[k + 128], e = src[k + 192];
        dst[k] = a; dst[k + 64] = b; dst[k + 128] = c; dst[k + 192] = e;
    }
    for (; k < count; k += 64) dst[k] = src[k];
}

__device__ __forceinline__ void copy_row(float *dst, const float *src, int count, int lane) {
    if ((count & 3) == 0 && ((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0) copy_vec((float4 *)dst, (const float4 *)src, count >> 2, lane);
    else copy_vec(dst, src, count, lane);
}

// rows r0 .. r0 + cnt - 1 of a field that is w floats wide: dst is contiguous over the tile, lane k takes element k of it
__device__ __forceinline__ void copy_narrow(float *dst, const float *src, int w, long long cell, int cnt, int lane) {
    const int total = cnt * w;
    for (int k0 = 0; k0 < total; k0 += 64) {   // every lane takes part in the shuffle
        const int k = k0 + lane;
        const int rr = min(k / w, cnt - 1);
        const long long c = __shfl(cell, rr);
        if (k < total) dst[k] = src[(size_t)c * w + (k - rr * w)];
    }
}

__device__ __forceinline__ void batch_cell(const FmarlBatchSrc &s, const int64_t *index, int64_t r, int mode, int chunk, int64_t chunks,
                                           long long &slot, long long &cell, int &agent) {
    const int64_t T = s.T, n = s.n, N = s.N;
    int64_t t, e, a;
    if (mode == 0) {
        const int64_t f = index[r];
        a = f % N; e = (f / N) % n; t = f / (N * n);
    } else {
        const int64_t l = r / chunks, j = r - l * chunks;
        const int64_t f = index[j] * chunk + l;
        t = f % T; a = (f / T) % N; e = f / (T * N);
    }
    slot = t * n + e; cell = slot * N + a; agent = (int)a;
}

constexpr int kNarrow = 16;   // fields up to this many floats per row go out tile-wise

// (not inlined: fifteen inlined copies of the two loops cost the kernel 179 VGPRs = two waves per SIMD; as a function the whole
// kernel needs a third of that, and a call per field and tile of 64 rows is nothing beside the rows it copies)
__device__ __attribute__((noinline)) void copy_field(float *dst, const float *src, int w, long long cell_l, int64_t r0, int cnt, int lane) {
    if (!dst) return;
    if (w <= kNarrow) { copy_narrow(dst + (size_t)r0 * w, src, w, cell_l, cnt, lane); return; }
    for (int i = 0; i < cnt; ++i) {
        const long long c = __shfl(cell_l, i);
        copy_row(dst + (size_t)(r0 + i) * w, src + (size_t)c * w, w, lane);
    }
}

__global__ __launch_bounds__(256) void minibatch_gather_kernel(FmarlBatchSrc s_arg, FmarlBatchDst d_arg, const int64_t *index, int64_t rows,
                                                               int mode, int chunk, int64_t chunks, int tile) {
    // the two pointer blocks (35 pointers) are read from the kernel's argument segment where a field is used, through a pointer the
    // compiler cannot see through: loaded up front they do not fit the scalar registers (286 of them were spilled into vector
    // lanes, every use a v_readlane)
    typedef const FmarlBatchSrc __attribute__((address_space(4))) *SrcArg;
    typedef const FmarlBatchDst __attribute__((address_space(4))) *DstArg;
    const uint64_t ka = pin_sgpr((uint64_t)__builtin_amdgcn_kernarg_segment_ptr());
    const FmarlBatchSrc &s = *(const FmarlBatchSrc *)(SrcArg)ka;
    const FmarlBatchDst &d = *(const FmarlBatchDst *)(DstArg)(ka + sizeof(FmarlBatchSrc));
    static_assert(sizeof(FmarlBatchSrc) % 8 == 0, "the second argument follows the first without padding");
    (void)s_arg; (void)d_arg;
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int N = s.N;
    // tile = rows per wave: 64 for a large minibatch, fewer for a small one so that its wide rows still spread over the chip
    for (int64_t r0 = wave0 * tile; r0 < rows; r0 += nwaves * tile) {
        const int cnt = (int)min((int64_t)tile, rows - r0);
        const int64_t r = r0 + lane;
        long long slot = 0, cell = 0;
        int agent = 0;
        if (lane < cnt) batch_cell(s, index, r, mode, chunk, chunks, slot, cell, agent);
        if (lane < cnt) {   // one value per row: coalesced over the tile
            if (d.agent_id) d.agent_id[r] = agent;
            if (d.value_preds) d.value_preds[r] = s.value_preds[cell];
            if (d.returns) d.returns[r] = s.returns[cell];
            if (d.masks) d.masks[r] = s.masks[cell];
            if (d.active_masks) d.active_masks[r] = s.active_masks[cell];
            if (d.adv_targ) d.adv_targ[r] = s.advantages[cell];
            if (d.env_slot) d.env_slot[r] = slot;
        }
        if (d.share_agent_id)   // rows 0 .. N-1 (graph_mpe_runner.py:479-484)
            for (int k = lane; k < cnt * N; k += 64) d.share_agent_id[(size_t)r0 * N + k] = k % N;
        copy_field(d.obs, s.obs, s.D, cell, r0, cnt, lane);
        copy_field(d.actions, s.actions, s.act_dim, cell, r0, cnt, lane);
        copy_field(d.old_action_log_probs, s.action_log_probs, s.act_dim, cell, r0, cnt, lane);
        copy_field(d.available_actions, s.available_actions, s.avail_dim, cell, r0, cnt, lane);
        copy_field(d.node_obs, s.node_obs, s.E * s.F, cell, r0, cnt, lane);
        copy_field(d.share_obs, s.obs, N * s.D, slot, r0, cnt, lane);   // all agents' obs of the env
        copy_field(d.adj, s.adj_env, s.E * s.E, slot, r0, cnt, lane);    // one matrix per env
        if (mode == 0) {
            copy_field(d.rnn_states, s.rnn_states, s.rnn_elems, cell, r0, cnt, lane);
            copy_field(d.rnn_states_critic, s.rnn_states_critic, s.rnn_elems, cell, r0, cnt, lane);
        }
    }
    if (mode == 1) {   // the chunks' initial recurrent states
        for (int64_t j0 = wave0 * tile; j0 < chunks; j0 += nwaves * tile) {
            const int cnt = (int)min((int64_t)tile, chunks - j0);
            long long slot = 0, cell = 0;
            int agent = 0;
            if (lane < cnt) batch_cell(s, index, j0 + lane, mode, chunk, chunks, slot, cell, agent);   // row l = 0 of chunk j
            copy_field(d.rnn_states, s.rnn_states, s.rnn_elems, cell, j0, cnt, lane);
            copy_field(d.rnn_states_critic, s.rnn_states_critic, s.rnn_elems, cell, j0, cnt, lane);
        }
    }
}

}  // namespace fmarl
