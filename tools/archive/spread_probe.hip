// measurement aid (GPU box): does a store stream run faster when the SAME number of bytes is spread over a larger footprint?
// One 160 GB buffer; a launch writes `bytes` (one step of cfg 3: 8.3 GB) as pieces of `piece` bytes, piece k at offset k * stride * piece:
// stride 1 = one contiguous 8.3 GB region (a time slot), stride 19 = the same bytes spread over the whole buffer (what a slot would
// be if the slots of a ring were interleaved piece by piece).  Workgroups take their pieces in scattered order, a wave streams its
// quarter of a piece (1 KiB per store instruction).  build + run on the box: hipcc -O3 --offload-arch=gfx950 tools/archive/spread_probe.hip -o /tmp/spread && /tmp/spread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ __launch_bounds__(256) void spread(float4 *dst, unsigned piece16, unsigned n_pieces, unsigned order, unsigned stride) {
    const unsigned k = (unsigned)(((unsigned long long)blockIdx.x * order) % n_pieces);
    float4 *p = dst + (size_t)k * stride * piece16;
    const unsigned per_wave = piece16 / 4, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (unsigned i = wave * per_wave + lane; i < (wave + 1) * per_wave; i += 64) p[i] = make_float4((float)i, 1.f, 2.f, (float)k);
}
static unsigned gcdu(unsigned a, unsigned b) { while (b) { unsigned r = a % b; a = b; b = r; } return a; }
int main() {
    const size_t total = (size_t)160 << 30, bytes = (size_t)8317 * 1000 * 1000;
    float4 *buf; CK(hipMalloc(&buf, total));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rnd = 0; rnd < 2; ++rnd)
    for (size_t piece : {(size_t)64 << 10, (size_t)2 << 20, (size_t)32 << 20}) {
        for (unsigned stride : {1u, 4u, 19u}) {
            const unsigned n = (unsigned)(bytes / piece);
            unsigned order = (unsigned)(n * 0.6180339887) | 1u; while (gcdu(order, n) != 1) order += 2;
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(spread, dim3(n), dim3(256), 0, 0, buf, (unsigned)(piece / 16), n, order, stride);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (rep && ms < best) best = ms;
            }
            printf("piece %8zu KB  stride %2u (footprint %6.1f GB): %.4f ms  %.3f TB/s\n", piece >> 10, stride, (double)n * stride * piece / 1e9, best, (double)n * piece / best / 1e9);
        }
    }
    return 0;
}
