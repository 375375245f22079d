// Where the ~0.3 s of a cold HIP start go (tool): hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/hip_init_probe.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <sys/time.h>
#include <cstdio>
static double now() { timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec * 1e3 + tv.tv_usec * 1e-3; }
__global__ void k_noop(int* p) { if (p) *p = 1; }
int main()
{
    double t = now(), t0 = t;
    auto lap = [&](const char* w) { const double n = now(); printf("%-28s %8.1f ms\n", w, n - t); t = n; };
    int n = 0; hipGetDeviceCount(&n); lap("hipGetDeviceCount");
    hipSetDevice(0); lap("hipSetDevice");
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); lap("hipGetDeviceProperties");
    hipStream_t s[6]; for (auto& x : s) hipStreamCreate(&x); lap("6 x hipStreamCreate");
    hipEvent_t e[32]; for (auto& x : e) hipEventCreate(&x); lap("32 x hipEventCreate");
    void* h = nullptr; hipHostMalloc(&h, 4096, hipHostMallocMapped | hipHostMallocCoherent); lap("hipHostMalloc 4 KB");
    void* d = nullptr; hipMalloc(&d, 64 << 20); lap("hipMalloc 64 MB");
    hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, s[0], (int*)d); hipStreamSynchronize(s[0]); lap("first kernel + sync");
    void* big = nullptr; hipMalloc(&big, 8ULL << 30); lap("hipMalloc 8 GB");
    void* big2 = nullptr; hipMalloc(&big2, 32ULL << 30); lap("hipMalloc 32 GB");
    hipFree(big2); lap("hipFree 32 GB");
    hipFree(big); hipFree(d); lap("hipFree rest");
    printf("total %.1f ms\n", now() - t0);
    return 0;
}
