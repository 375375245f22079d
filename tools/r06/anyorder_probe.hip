// anyorder_probe.hip - round 6, question 1: can a kernel start while the previous kernel OF THE SAME STREAM is still draining?
//   (a) hipExtLaunchKernelGGL(..., hipExtAnyOrderLaunch): the AQL packet without its barrier bit (hip_ext.h says "not supported on GFX9xx" - measured here)
//   (b) ONE kernel whose later workgroups wait for earlier ones (in-order dispatch): start order per XCC, spin-wait on a lower workgroup id
// Every wait has a time limit (50 ms): a wrong assumption ends in an error count, not in a hung device.
// build: hipcc --offload-arch=gfx950 -O2 -o anyorder_probe anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline u64 now() { return wall_clock64(); }      // 100 MHz

__global__ void k_busy(u64* t0, u64* t1, int* flag, int iters, int slow_from)
{
    const int b = blockIdx.x;
    if (threadIdx.x == 0) t0[b] = now();
    // a dependent chain: the last workgroups (>= slow_from) run 4 x longer - a launch with a tail
    unsigned v = b * 2654435761u + threadIdx.x;
    const int n = b >= slow_from ? iters * 4 : iters;
    for (int i = 0; i < n; ++i) v = v * 1664525u + 1013904223u;
    if (v == 12345u) t1[b] = 1;
    __syncthreads();
    if (threadIdx.x == 0) { t1[b] = now(); __threadfence(); __hip_atomic_store(&flag[b], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
}

__global__ void k_dep(u64* t0, u64* t1, const int* flag, int nflag, int* err)
{
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        t0[b] = now();
        const int need = b % nflag;
        const u64 lim = t0[b] + 5000000ULL;        // 50 ms
        while (__hip_atomic_load(&flag[need], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(8);
            if (now() > lim) { atomicAdd(err, 1); break; }
        }
        t1[b] = now();
    }
}

// (b) one kernel: workgroup i >= half waits for workgroup i - half
__global__ void k_chain(u64* t0, u64* t1, int* flag, int half, int iters, int* err, unsigned* xcc)
{
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        t0[b] = now();
        unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[b] = id & 15u;
        if (b >= half) {
            const u64 lim = t0[b] + 5000000ULL;
            while (__hip_atomic_load(&flag[b - half], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(8);
                if (now() > lim) { atomicAdd(err, 1); break; }
            }
        }
    }
    __syncthreads();
    unsigned v = b * 2654435761u + threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1664525u + 1013904223u;
    if (v == 12345u) t1[b] = 1;
    __syncthreads();
    if (threadIdx.x == 0) { t1[b] = now(); __threadfence(); __hip_atomic_store(&flag[b], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
}

int main()
{
    const int G1 = 256 * 8 * 3 + 512, G2 = 4096, iters = 200000;      // ~3.25 waves of 256-thread workgroups at 8 per CU
    u64 *t0a, *t1a, *t0b, *t1b; int *flag, *err; unsigned* xcc;
    CK(hipMalloc(&t0a, G1 * 8)); CK(hipMalloc(&t1a, G1 * 8)); CK(hipMalloc(&t0b, 65536 * 8)); CK(hipMalloc(&t1b, 65536 * 8));
    CK(hipMalloc(&flag, 65536 * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&xcc, 65536 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<u64> h0a(G1), h1a(G1), h0b(65536), h1b(65536);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(flag, 0, 65536 * 4, s)); CK(hipMemsetAsync(err, 0, 4, s));
            hipLaunchKernelGGL(k_busy, dim3(G1), dim3(256), 0, s, t0a, t1a, flag, iters, G1 - 64);
            if (mode == 0) hipLaunchKernelGGL(k_dep, dim3(G2), dim3(64), 0, s, t0b, t1b, (const int*)flag, G1, err);
            else hipExtLaunchKernelGGL(k_dep, dim3(G2), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, t0b, t1b, (const int*)flag, G1, err);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(s));
            int herr = 0;
            CK(hipMemcpy(h0a.data(), t0a, G1 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1a.data(), t1a, G1 * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h0b.data(), t0b, G2 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1b.data(), t1b, G2 * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            const u64 a0 = *std::min_element(h0a.begin(), h0a.end()), a1 = *std::max_element(h1a.begin(), h1a.end());
            std::vector<u64> ends(h1a); std::sort(ends.begin(), ends.end());
            const u64 b0 = *std::min_element(h0b.begin(), h0b.begin() + G2), b1 = *std::max_element(h1b.begin(), h1b.begin() + G2);
            int early = 0; for (int i = 0; i < G2; ++i) early += h0b[i] < a1;
            printf("%s rep %d: first kernel %.1f us (90 %% of its workgroups done at %.1f us), dependent kernel starts at %.1f us, ends at %.1f us; %d of %d dependent workgroups started before the first kernel ended; timeouts %d\n",
                   mode ? "any-order" : "in-order ", rep, (a1 - a0) / 100.0, (ends[G1 * 9 / 10] - a0) / 100.0, (b0 - a0) / 100.0, (b1 - a0) / 100.0, early, G2, herr);
        }
    }
    // (b) one kernel, later workgroups depend on earlier ones
    for (int rep = 0; rep < 2; ++rep) {
        const int half = 20000, G = 2 * half;
        CK(hipMemsetAsync(flag, 0, 65536 * 4, s)); CK(hipMemsetAsync(err, 0, 4, s));
        hipLaunchKernelGGL(k_chain, dim3(G), dim3(256), 0, s, t0b, t1b, flag, half, 20000, err, xcc);
        CK(hipGetLastError()); CK(hipStreamSynchronize(s));
        std::vector<unsigned> hx(G); int herr = 0;
        CK(hipMemcpy(h0b.data(), t0b, G * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1b.data(), t1b, G * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        // start order: per XCC, is start time monotone in the workgroup id?  how far ahead does any workgroup start of a lower id (globally)?
        long inv_x = 0, inv_g = 0; u64 last[16] = {0}; u64 gl = 0; int xmod = 0;
        for (int i = 0; i < G; ++i) {
            const unsigned x = hx[i] & 15u; xmod += (int)(x == (unsigned)(i % 8));
            if (h0b[i] + 20 < last[x]) ++inv_x;       // 0.2 us slack
            last[x] = std::max(last[x], h0b[i]);
            if (h0b[i] + 20 < gl) ++inv_g; gl = std::max(gl, h0b[i]);
        }
        const u64 a0 = *std::min_element(h0b.begin(), h0b.begin() + G), a1 = *std::max_element(h1b.begin(), h1b.begin() + G);
        printf("one kernel, %d workgroups, second half waits for the first: %.1f us, timeouts %d; workgroups on xcc == id %% 8: %d; start-order inversions per xcc %ld, globally %ld\n",
               G, (a1 - a0) / 100.0, herr, xmod, inv_x, inv_g);
    }
    return 0;
}
