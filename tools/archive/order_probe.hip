// measurement aid (GPU box): which ORDER of chunks lets a pure store stream run fastest?  One 160 GB buffer, a launch writes it
// whole (footprint = everything) in 64 KB ... 1 MB chunks; workgroup b takes chunk f(b):
//   0 dispatch order      f(b) = b
//   1 golden section      f(b) = b * order mod n                       (what env_block / fmarl_store_stream do)
//   2 XCD regions         workgroups are dealt round-robin to the 8 XCDs: XCD x (= b mod 8) walks its own eighth of the buffer
//   3 XCD regions, golden the same with the golden-section order inside the eighth
//   4 bit reversal        f(b) = bit-reversed b (n a power of two)
// and with `persist` workgroups that live for the whole launch (0 = one workgroup per chunk).
// build + run on the box: hipcc -O3 --offload-arch=gfx950 tools/archive/order_probe.hip -o /tmp/order && /tmp/order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ unsigned map_chunk(unsigned c, unsigned n, unsigned order, unsigned order8, int mode, int bits) {
    if (mode == 1) return (unsigned)(((unsigned long long)c * order) % n);
    if (mode == 2) return (c & 7u) * (n >> 3) + (c >> 3);
    if (mode == 3) return (c & 7u) * (n >> 3) + (unsigned)(((unsigned long long)(c >> 3) * order8) % (n >> 3));
    if (mode == 4) return __brev(c) >> (32 - bits);
    return c;
}
__global__ __launch_bounds__(256) void stream(float4 *dst, unsigned chunk16, unsigned n, unsigned order, unsigned order8, int mode, int bits) {
    for (unsigned c = blockIdx.x; c < n; c += gridDim.x) {
        const unsigned k = map_chunk(c, n, order, order8, mode, bits);
        float4 *p = dst + (size_t)k * chunk16;
        const unsigned per_wave = chunk16 / 4, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (unsigned i = wave * per_wave + lane; i < (wave + 1) * per_wave; i += 64) p[i] = make_float4((float)i, 1.f, 2.f, (float)k);
    }
}
static unsigned gcdu(unsigned a, unsigned b) { while (b) { unsigned r = a % b; a = b; b = r; } return a; }
static unsigned golden(unsigned n) { unsigned o = (unsigned)(n * 0.6180339887) | 1u; while (gcdu(o, n) != 1) o += 2; return o; }
int main() {
    const size_t total = (size_t)128 << 30;   // a power of two: bit reversal works
    float4 *buf; CK(hipMalloc(&buf, total));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char *names[] = {"dispatch order", "golden section", "XCD regions", "XCD regions, golden", "bit reversal"};
    for (size_t chunk : {(size_t)64 << 10, (size_t)256 << 10, (size_t)1 << 20}) {
        const unsigned n = (unsigned)(total / chunk);
        int bits = 0; while ((1u << bits) < n) ++bits;
        for (int mode = 0; mode < 5; ++mode)
            for (unsigned persist : {0u, 1024u, 2048u}) {
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(a));
                    hipLaunchKernelGGL(stream, dim3(persist ? persist : n), dim3(256), 0, 0, buf, (unsigned)(chunk / 16), n, golden(n), golden(n >> 3), mode, bits);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    if (rep && ms < best) best = ms;
                }
                printf("chunk %5zu KB  %-20s persist %4u: %8.3f ms  %.3f TB/s\n", chunk >> 10, names[mode], persist, best, (double)total / best / 1e9);
                fflush(stdout);
            }
    }
    return 0;
}
