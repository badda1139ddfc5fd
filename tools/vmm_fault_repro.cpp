// measurement aid (GPU box), torch-free: the two faults of re-used virtual address ranges that fmarl_ring_alloc works around, on the bare
// HIP virtual-memory calls (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess / hipMemUnmap / hipMemRelease /
// hipMemAddressFree) with a fill kernel and a check kernel -- no PyTorch, no libfmarl.  Three policies over the same sequence of arrays
// (sizes alternate, each array = 2 slots of 16 MiB pieces, virtual piece j of slot t <- physical piece j * 2 + t, as the library maps):
//   C  control   every array gets a NEW reservation, freed arrays keep theirs (what libfmarl does): expected 0 wrong words
//   A  re-reserve  a freed array's range goes back to the runtime (hipMemAddressFree) and the next hipMemAddressReserve may hand the
//                same addresses out again (round 4's fault)
//   B  kept range  ONE reservation as large as the largest array; every array maps fresh pieces at its start (round 5's fault)
// Every array: kernel fill with a pattern of (array index, word index) -> device synchronize -> kernel check at once, after 0.3 s and
// after 2 s (round 4 saw words change seconds later) -> hipMemcpy of the first piece to the host and a host check -> unmap, release.
//   hipcc -O2 --offload-arch=gfx950 -o tools/vmm_fault_repro tools/vmm_fault_repro.cpp && tools/vmm_fault_repro [arrays=8]
// Output: one line per array and policy; the summary line per policy is what profiles/r6_vmm_fault_repro.txt keeps.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define OK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } \
    } while (0)

__global__ void fill_kernel(uint4 *dst, size_t n16, uint32_t salt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint32_t w = (uint32_t)i * 2654435761u ^ salt;
        dst[i] = make_uint4(w, ~w, w + 1u, salt);
    }
}
__global__ void check_kernel(const uint4 *src, size_t n16, uint32_t salt, unsigned long long *bad, unsigned long long *zero) {
    unsigned long long mine = 0, z = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint32_t w = (uint32_t)i * 2654435761u ^ salt;
        const uint4 v = src[i];
        const int b = (v.x != w) + (v.y != ~w) + (v.z != w + 1u) + (v.w != salt);
        mine += b;
        if (b && v.x == 0 && v.y == 0 && v.z == 0 && v.w == 0) z += 4;
    }
    if (mine) atomicAdd(bad, mine);
    if (z) atomicAdd(zero, z);
}

static const size_t kPiece = (size_t)16 << 20;
static hipMemAllocationProp g_prop;
static int g_dev = 0;
static unsigned long long *g_counts = nullptr;   // device: [bad, zero]

struct Array {
    void *ptr = nullptr;
    size_t total = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

static void map_pieces(Array &a, void *at, size_t total) {
    a.ptr = at; a.total = total;
    const size_t count = total / kPiece, slots = 2, per_slot = count / slots;
    a.handles.resize(count);
    for (size_t k = 0; k < count; ++k) OK(hipMemCreate(&a.handles[k], kPiece, &g_prop, 0));
    for (size_t t = 0; t < slots; ++t)
        for (size_t j = 0; j < per_slot; ++j) OK(hipMemMap((char *)at + (t * per_slot + j) * kPiece, kPiece, 0, a.handles[j * slots + t], 0));
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = g_dev; acc.flags = hipMemAccessFlagsProtReadWrite;
    OK(hipMemSetAccess(at, total, &acc, 1));
}
static void unmap_pieces(Array &a) {
    OK(hipDeviceSynchronize());
    for (size_t k = 0; k < a.handles.size(); ++k) OK(hipMemUnmap((char *)a.ptr + k * kPiece, kPiece));
    for (auto h : a.handles) OK(hipMemRelease(h));
    a.handles.clear();
}
static void check(const Array &a, uint32_t salt, unsigned long long out[2]) {
    OK(hipMemset(g_counts, 0, 2 * sizeof(*g_counts)));
    hipLaunchKernelGGL(check_kernel, dim3(2048), dim3(256), 0, 0, (const uint4 *)a.ptr, a.total / 16, salt, g_counts, g_counts + 1);
    OK(hipMemcpy(out, g_counts, 2 * sizeof(*g_counts), hipMemcpyDeviceToHost));
}

// fill array `a` by a kernel, then look at it four ways; -> wrong words seen by the worst kernel read
static unsigned long long exercise(const char *policy, int k, Array &a, bool same_address) {
    const uint32_t salt = 0x9e3779b9u * (uint32_t)(k + 1) ^ (uint32_t)policy[0];
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (uint4 *)a.ptr, a.total / 16, salt);
    OK(hipDeviceSynchronize());
    unsigned long long now[2], later[2], last[2];
    check(a, salt, now);
    std::this_thread::sleep_for(std::chrono::milliseconds(300));
    check(a, salt, later);
    std::this_thread::sleep_for(std::chrono::milliseconds(2000));
    check(a, salt, last);
    std::vector<uint32_t> host(kPiece / 4);
    OK(hipMemcpy(host.data(), a.ptr, kPiece, hipMemcpyDeviceToHost));   // the copy engines' view of the first piece
    unsigned long long host_bad = 0;
    for (size_t i = 0; i < kPiece / 16; ++i) {
        const uint32_t w = (uint32_t)i * 2654435761u ^ salt;
        host_bad += (host[4 * i] != w) + (host[4 * i + 1] != ~w) + (host[4 * i + 2] != w + 1u) + (host[4 * i + 3] != salt);
    }
    const unsigned long long words = a.total / 4;
    printf("%s array %d: %4zu MiB at %p%s  wrong words by kernel reads: at once %llu (%llu of them zero), after 0.3 s %llu, after 2.3 s %llu, of %llu; "
           "first piece copied to the host: %llu wrong\n", policy, k, a.total >> 20, a.ptr, same_address ? " (an address used before)" : "",
           now[0], now[1], later[0], last[0], words, host_bad);
    fflush(stdout);
    unsigned long long worst = now[0] > later[0] ? now[0] : later[0];
    return worst > last[0] ? worst : last[0];
}

int main(int argc, char **argv) {
    const int arrays = argc > 1 ? atoi(argv[1]) : 8;
    const size_t sizes[] = {(size_t)480 << 20, (size_t)96 << 20, (size_t)480 << 20, (size_t)96 << 20, (size_t)256 << 20};
    OK(hipGetDevice(&g_dev));
    int vmm = 0;
    OK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, g_dev));
    if (!vmm) { printf("no virtual memory management on this device\n"); return 1; }
    g_prop = {};
    g_prop.type = hipMemAllocationTypePinned; g_prop.location.type = hipMemLocationTypeDevice; g_prop.location.id = g_dev;
    size_t gran = 0;
    OK(hipMemGetAllocationGranularity(&gran, &g_prop, hipMemAllocationGranularityMinimum));
    hipDeviceProp_t dp;
    OK(hipGetDeviceProperties(&dp, g_dev));
    int rt = 0;
    OK(hipRuntimeGetVersion(&rt));
    printf("device %s (%s), HIP runtime %d, allocation granularity %zu, pieces of %zu MiB, %d arrays per policy\n", dp.name, dp.gcnArchName, rt, gran,
           kPiece >> 20, arrays);
    OK(hipMalloc((void **)&g_counts, 2 * sizeof(*g_counts)));

    unsigned long long totals[3] = {0, 0, 0};
    int faulty[3] = {0, 0, 0};
    // ---- C: a new reservation per array, freed arrays keep theirs
    {
        std::vector<void *> kept;
        for (int k = 0; k < arrays; ++k) {
            const size_t total = sizes[k % 5];
            void *at = nullptr;
            OK(hipMemAddressReserve(&at, total, (size_t)2 << 20, nullptr, 0));
            Array a;
            map_pieces(a, at, total);
            const unsigned long long bad = exercise("C control   ", k, a, false);
            totals[0] += bad; faulty[0] += bad != 0;
            unmap_pieces(a);
            kept.push_back(at);   // never freed, never used again
        }
    }
    // ---- B: one kept reservation, fresh pieces per array
    {
        void *at = nullptr;
        OK(hipMemAddressReserve(&at, (size_t)480 << 20, (size_t)2 << 20, nullptr, 0));
        for (int k = 0; k < arrays; ++k) {
            Array a;
            map_pieces(a, at, sizes[k % 5]);
            const unsigned long long bad = exercise("B kept range", k, a, k > 0);
            totals[2] += bad; faulty[2] += bad != 0;
            unmap_pieces(a);
        }
    }
    // ---- A: the range goes back to the runtime after every array
    {
        std::vector<void *> seen;
        for (int k = 0; k < arrays; ++k) {
            const size_t total = sizes[k % 5];
            void *at = nullptr;
            OK(hipMemAddressReserve(&at, total, (size_t)2 << 20, nullptr, 0));
            bool again = false;
            for (void *s : seen) again |= ((char *)at < (char *)s + ((size_t)480 << 20) && (char *)s < (char *)at + total);
            Array a;
            map_pieces(a, at, total);
            const unsigned long long bad = exercise("A re-reserve", k, a, again);
            totals[1] += bad; faulty[1] += bad != 0;
            unmap_pieces(a);
            OK(hipMemAddressFree(at, total));
            seen.push_back(at);
        }
    }
    printf("SUMMARY C control (new range per array, freed ranges kept idle): %d of %d arrays faulty, %llu wrong words\n", faulty[0], arrays, totals[0]);
    printf("SUMMARY A re-reserve (range returned to the runtime, reserved again): %d of %d arrays faulty, %llu wrong words\n", faulty[1], arrays, totals[1]);
    printf("SUMMARY B kept range (fresh pieces mapped into one kept range): %d of %d arrays faulty, %llu wrong words\n", faulty[2], arrays, totals[2]);
    return 0;
}
