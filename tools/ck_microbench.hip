// ck_microbench.hip - the list-A checkpoint pass (k_myers_ck<8, 16, true>, ext_rcwalk.h) alone, on random full blocks: per-launch time
// at three list sizes (one wave per SIMD, one full round of 8 waves per SIMD, the biggest round of the bench) for the variants the
// kernel's `flags` select.  The SHW pass has no data-dependent control flow, so random words time like real ones.
// (The structural variants of round 4 - half the windows, no post-pass, no windows, no work counters: -DNECAT_CK_MV=5 .. 10 - were cut out of the product header in
// round 5; they are at commit 240f31f, their numbers in profiles/NOTES_r04.md 10.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-promote-alloca-to-lds -I necat_amd/csrc -o tools/ck_microbench tools/ck_microbench.hip
#include <algorithm>
#include <chrono>
#include <mutex>
#include <unordered_map>
#include <numeric>
#define NECAT_CK_MICRO 1
#include "runtime.h"
#include "ext_kernels.h"
#include "ext_tail.h"
#include "ext_rcwalk.h"
using namespace necat;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill(u64* p, size_t n, u64 seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u64 x = (i + 1) * 0x9E3779B97F4A7C15ULL ^ seed; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
        p[i] = x;
    }
}

int main(int argc, char** argv)
{
    const u32 sizes[3] = {8192u, 65536u, 229376u};
    const u32 nmax = sizes[2];
    constexpr int NW = 8, TW = 16, FW = 2 * NW + TW, G = 8;
    u64* frag; ulonglong2* ck; u64* hc; BlockResult* res; unsigned long long* stats; u32* ndev;
    CHECK(hipMalloc(&frag, (size_t)nmax * FW * 8));
    CHECK(hipMalloc(&ck, (size_t)nmax * kRcCk16 * G * 16));
    CHECK(hipMalloc(&hc, (size_t)nmax * kRcCk * G * 8));
    CHECK(hipMalloc(&res, (size_t)nmax * sizeof(BlockResult)));
    CHECK(hipMalloc(&stats, kStatBytes)); CHECK(hipMemset(stats, 0, kStatBytes));
    CHECK(hipMalloc(&ndev, 16));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, frag, (size_t)nmax * FW, 12345ULL);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void*)k_myers_ck<NW, TW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
    const u32 variants[3] = {0u, 1u << 24, 1u << 20};
    const char* vname[3] = {"post-pass minimum, unrolled windows", "minimum inside the pass (round-3 loop)", "post-pass, NO checkpoint / delta stores"};
    // occupancy: dynamic LDS per one-wave workgroup caps the waves a CU holds (160 KB / (1 KB + lds))
    const u32 lds_list[4] = {0u, argc > 1 ? 0u : 9u << 10, 19u << 10, 39u << 10};       // 32 (8 per SIMD), 16, 8, 4 waves per CU
    for (int li = 1; li < (argc > 1 ? 2 : 4); ++li) {
        const u32 n = sizes[2];
        const u32 cnt[4] = {n, 0, 0, 0};
        CHECK(hipMemcpy(ndev, cnt, 16, hipMemcpyHostToDevice));
        for (int v = 0; v < 2; ++v) {
            float best = 1e9f;
            for (int r = 0; r < 4; ++r) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL((k_myers_ck<NW, TW, true>), dim3((n + 7) / 8), dim3(64), lds_list[li], 0, (const BlockItem*)nullptr, (const u32*)ndev, n, (const u64*)frag, ck, hc, 0.5,
                                   res, stats, 1 << 20, 0u, n, variants[v] | 1u);
                CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r) best = std::min(best, ms);
            }
            printf("%7u blocks, %2u waves per CU | %-40s | best %8.1f us | %.2f ns per block\n", n, 160u / (1u + (lds_list[li] >> 10)), vname[v], best * 1e3, best * 1e6 / n);
        }
    }
    for (int si = 0; si < (argc > 1 ? 0 : 3); ++si) {
        const u32 n = sizes[si];
        const u32 cnt[4] = {n, 0, 0, 0};
        CHECK(hipMemcpy(ndev, cnt, 16, hipMemcpyHostToDevice));
        for (int v = 0; v < 3; ++v) {
            float best = 1e9f, sum = 0;
            const int reps = 6;
            for (int r = 0; r < reps + 1; ++r) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL((k_myers_ck<NW, TW, true>), dim3((n + 7) / 8), dim3(64), 0, 0, (const BlockItem*)nullptr, (const u32*)ndev, n, (const u64*)frag, ck, hc, 0.5,
                                   res, stats, 1 << 20, 0u, n, variants[v] | 1u);
                CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r) { best = std::min(best, ms); sum += ms; }
            }
            std::vector<BlockResult> h(4);
            CHECK(hipMemcpy(h.data(), res, sizeof(BlockResult) * 4, hipMemcpyDeviceToHost));
            printf("%7u blocks (%5u waves) | %-40s | best %8.1f us, mean %8.1f us | %.1f ns per block | block 0: dist %d endc %d\n", n, (n + 7) / 8, vname[v], best * 1e3, sum / reps * 1e3,
                   best * 1e6 / n, h[0].dist, h[0].endc);
        }
    }
    return 0;
}
