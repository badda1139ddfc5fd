// Write-bandwidth microbenchmark for MI355X: what a store-only stream of the step kernel's shape can reach.
// Patterns: 0 = flat grid-stride float4 stream; 1 = lane owns a column chunk and walks 32 rows 3168 B apart
// (the emit_graph pattern); 2 = wave owns a row and streams it.  Usage: wbench [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void flat(float4 *dst, size_t n4) {
    float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = v;
}
// one block = 8 envs x (32 rows x 198 chunks): item = (env, chunk), inner loop over rows
__global__ __launch_bounds__(256) void colwalk(float4 *dst, int nenv_total) {
    const int C4 = 198, N = 32;
    int env0 = blockIdx.x * 8;
    for (int it = threadIdx.x; it < 8 * C4; it += 256) {
        int el = it / C4, c = it - el * C4;
        float4 *d = dst + ((size_t)(env0 + el) * N) * C4 + c;
        float4 v = make_float4(1.f, 2.f, 3.f, (float)c);
        for (int i = 0; i < N; ++i) { v.x += 1.f; d[(size_t)i * C4] = v; }
    }
}
// wave owns a row: rows of the block = 8 * 32 = 256 rows, 4 waves -> 64 rows per wave, row = 198 chunks
__global__ __launch_bounds__(256) void rowstream(float4 *dst, int nenv_total) {
    const int C4 = 198, N = 32;
    int env0 = blockIdx.x * 8;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int row = wave; row < 8 * N; row += 4) {
        float4 *d = dst + ((size_t)env0 * N + row) * C4;
        float4 v = make_float4(1.f, 2.f, 3.f, (float)row);
        for (int c = lane; c < C4; c += 64) d[c] = v;
    }
}
typedef float floatx4 __attribute__((ext_vector_type(4)));
// env-per-wave row streaming with non-temporal stores
__global__ __launch_bounds__(256) void rowstream_nt(float4 *dst, int nenv_total) {
    const int C4 = 198, N = 32;
    int env0 = blockIdx.x * 8;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int el = wave; el < 8; el += 4) {
        float4 *d = dst + ((size_t)(env0 + el) * N) * C4;
        for (int row = 0; row < N; ++row, d += C4) {
            float4 v = make_float4(1.f, 2.f, 3.f, (float)row);
            floatx4 w = {v.x, v.y, v.z, v.w};
            for (int c = lane; c < C4; c += 64) __builtin_nontemporal_store(w, (floatx4 *)&d[c]);
        }
    }
}
// wave owns an env and streams its 32 rows (the emit_node_rows order), plain stores
__global__ __launch_bounds__(256) void envwave(float4 *dst, int nenv_total) {
    const int C4 = 198, N = 32;
    int env0 = blockIdx.x * 8;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int el = wave; el < 8; el += 4) {
        float4 *d = dst + ((size_t)(env0 + el) * N) * C4;
        for (int row = 0; row < N; ++row, d += C4) {
            float4 v = make_float4(1.f, 2.f, 3.f, (float)row);
            for (int c = lane; c < C4; c += 64) d[c] = v;
        }
    }
}
// fully linear per block: block's 1 MB region streamed by all 256 threads
__global__ __launch_bounds__(256) void blocklinear(float4 *dst, int nenv_total) {
    const size_t per_block = (size_t)8 * 32 * 198;
    float4 *d = dst + (size_t)blockIdx.x * per_block;
    float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (size_t i = threadIdx.x; i < per_block; i += 256) d[i] = v;
}
int main(int argc, char **argv) {
    const int nenv = 65536;
    const size_t n4 = (size_t)nenv * 32 * 198;  // float4 count = 6.64 GB
    float4 *buf; CK(hipMalloc(&buf, n4 * 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int pat = 0; pat < 7; ++pat) {
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(a));
            if (pat == 0) hipLaunchKernelGGL(flat, dim3(2048), dim3(256), 0, 0, buf, n4);
            if (pat == 1) hipLaunchKernelGGL(colwalk, dim3(nenv / 8), dim3(256), 0, 0, buf, nenv);
            if (pat == 2) hipLaunchKernelGGL(rowstream, dim3(nenv / 8), dim3(256), 0, 0, buf, nenv);
            if (pat == 3) hipLaunchKernelGGL(blocklinear, dim3(nenv / 8), dim3(256), 0, 0, buf, nenv);
            if (pat == 4) hipLaunchKernelGGL(flat, dim3(256 * 8 * 4), dim3(256), 0, 0, buf, n4);
            if (pat == 5) hipLaunchKernelGGL(envwave, dim3(nenv / 8), dim3(256), 0, 0, buf, nenv);
            if (pat == 6) hipLaunchKernelGGL(rowstream_nt, dim3(nenv / 8), dim3(256), 0, 0, buf, nenv);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (rep && ms < best) best = ms;
        }
        printf("pattern %d: %.3f ms  %.2f TB/s\n", pat, best, n4 * 16.0 / best / 1e9);
    }
    return 0;
}
