// rcwalk_microbench.hip - the recomputing walk of list A (k_rcwalk3<8, 16, 512, 1024>, ext_rcwalk3.h, and its predecessor k_rcwalk2w, ext_rcwalk.h) alone, on synthetic 512 x 512 blocks
// (random query, target = the query with 12 % substitutions / insertions / deletions) whose checkpoints, deltas and results come from
// k_myers_ck on the same fragments.  Per-launch time by list size, by workgroups per CU (dynamic LDS), with / without kept ops (the
// `found` flag of the block's task), with raised wave priority; with -DNECAT_RC_TIMING also with the walk / the recompute of every segment
// done twice (what each phase costs = the difference).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-promote-alloca-to-lds -I necat_amd/csrc -o tools/rcwalk_microbench tools/rcwalk_microbench.hip
#include <algorithm>
#include <chrono>
#include <mutex>
#include <random>
#include <unordered_map>
#include <numeric>
#include "runtime.h"
#include "ext_kernels.h"
#include "ext_tail.h"
#include "ext_rcwalk.h"
#include "ext_rcwalk3.h"
using namespace necat;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    constexpr int NW = 8, TW = 16, FW = 2 * NW + TW, G = 8, N = 512, MAXOPS = 1024;
    const u32 nmax = 221184;                       // 3456 workgroups of 64 blocks: the size of the bench's big rounds
    const double err = argc > 1 ? atof(argv[1]) : 0.12;
    const int store_pct = argc > 2 ? atoi(argv[2]) : 0;
    const bool store_sorted = argc > 3 && atoi(argv[3]);   // the blocks that keep their ops at the back of the list instead of scattered over it          // per cent of the blocks whose task keeps its ops in the "mixed" runs (task 1: found = 0)
    std::vector<u64> hfrag((size_t)nmax * FW, 0);
    {
        std::mt19937_64 rng(12345);
        std::uniform_real_distribution<double> U(0.0, 1.0);
        std::vector<int> q(N), t(N);
        for (u32 x = 0; x < nmax; ++x) {
            for (int i = 0; i < N; ++i) q[i] = (int)(rng() & 3);
            int qi = 0, ti = 0;
            while (ti < N) {
                if (qi >= N) { t[ti++] = (int)(rng() & 3); continue; }
                const double r = U(rng);
                if (r < err / 3) { t[ti++] = (q[qi] + 1 + (int)(rng() % 3)) & 3; ++qi; }      // substitution
                else if (r < 2 * err / 3) t[ti++] = (int)(rng() & 3);                         // insertion into the target
                else if (r < err) ++qi;                                                       // deletion
                else t[ti++] = q[qi++];
            }
            u64* dst = hfrag.data() + (size_t)(x >> 6) * FW * 64 + (x & 63);
            for (int ch = 0; ch < NW; ++ch) {
                u64 lo = 0, hi = 0;
                for (int i = 0; i < 64; ++i) { lo |= (u64)(q[ch * 64 + i] & 1) << i; hi |= (u64)((q[ch * 64 + i] >> 1) & 1) << i; }
                dst[(size_t)ch * 64] = ~lo; dst[(size_t)(NW + ch) * 64] = ~hi;
            }
            for (int w = 0; w < TW; ++w) {
                u64 v = 0;
                for (int i = 0; i < 32; ++i) v |= (u64)t[w * 32 + i] << (2 * i);
                dst[(size_t)(2 * NW + w) * 64] = v;
            }
        }
    }
    u64* frag; ulonglong2* ck; u64* hc; BlockResult* res; unsigned long long* stats; u32* ndev; BlockItem* items; ExtTask* tasks; WalkOut* wout; u8* ops; int* errf;
    CHECK(hipMalloc(&frag, hfrag.size() * 8)); CHECK(hipMemcpy(frag, hfrag.data(), hfrag.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&ck, (size_t)nmax * kRcCk16 * G * 16));
    CHECK(hipMalloc(&hc, (size_t)nmax * kRcCk * G * 8));
    CHECK(hipMalloc(&res, (size_t)nmax * sizeof(BlockResult)));
    CHECK(hipMalloc(&stats, kStatBytes)); CHECK(hipMemset(stats, 0, kStatBytes));
    CHECK(hipMalloc(&ndev, 16)); CHECK(hipMalloc(&errf, 4)); CHECK(hipMemset(errf, 0, 4));
    CHECK(hipMalloc(&wout, (size_t)nmax * sizeof(WalkOut)));
    CHECK(hipMalloc(&ops, (size_t)(nmax / 64) * MAXOPS * 64));
    {
        std::vector<BlockItem> hi(nmax);
        for (u32 x = 0; x < nmax; ++x) { memset(&hi[x], 0, sizeof(BlockItem)); hi[x].task = (store_sorted ? (x >= (u32)((u64)nmax * (100 - store_pct) / 100)) : (((x * 2654435761u) >> 8) % 100u < (u32)store_pct)) ? 1 : 0; hi[x].qn = (i16)N; hi[x].tn = (i16)N; }
        CHECK(hipMalloc(&items, (size_t)nmax * sizeof(BlockItem))); CHECK(hipMemcpy(items, hi.data(), (size_t)nmax * sizeof(BlockItem), hipMemcpyHostToDevice));
        CHECK(hipMalloc(&tasks, 2 * sizeof(ExtTask)));
    }
    const u32 cnt[4] = {nmax, 0, 0, 0};
    CHECK(hipMemcpy(ndev, cnt, 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_myers_ck<NW, TW, true>), dim3((nmax + 7) / 8), dim3(64), 0, 0, (const BlockItem*)nullptr, (const u32*)ndev, nmax, (const u64*)frag, ck, hc, 0.5, res, stats, 1 << 20, 0u, nmax, 1u);
    CHECK(hipDeviceSynchronize());
    {
        std::vector<BlockResult> h(4096);
        CHECK(hipMemcpy(h.data(), res, sizeof(BlockResult) * h.size(), hipMemcpyDeviceToHost));
        double sd = 0, se = 0; int bad = 0;
        for (auto& r : h) { if (r.dist < 0) ++bad; else { sd += r.dist; se += r.endc; } }
        printf("blocks: mean distance %.1f, mean end column %.1f, %d without an alignment (of %zu)\n", sd / (h.size() - bad), se / (h.size() - bad), bad, h.size());
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void*)k_rcwalk2w<NW, TW, N, MAXOPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
    CHECK(hipFuncSetAttribute((const void*)k_rcwalk3<NW, TW, N, MAXOPS, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
    CHECK(hipFuncSetAttribute((const void*)k_rcwalk3<NW, TW, N, MAXOPS, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
    int kern = 4;            // 3: k_rcwalk3 on 32-diagonal records, 4: on 16-diagonal records
    auto launch3 = [&](u32 n, u32 lds, u8* o, WalkOut* w, u32 opts) {
        if (kern == 4) hipLaunchKernelGGL((k_rcwalk3<NW, TW, N, MAXOPS, 16>), dim3((n + 63) / 64), dim3(128), lds, 0, (const BlockItem*)items, n, (const u32*)ndev, n, (const u64*)frag, (const ulonglong2*)ck,
                                          (const u64*)hc, (const BlockResult*)res, (const ExtTask*)tasks, 0, 8, o, w, stats, errf, 1u, 0u, n, opts);
        else hipLaunchKernelGGL((k_rcwalk3<NW, TW, N, MAXOPS, 32>), dim3((n + 63) / 64), dim3(128), lds, 0, (const BlockItem*)items, n, (const u32*)ndev, n, (const u64*)frag, (const ulonglong2*)ck,
                                (const u64*)hc, (const BlockResult*)res, (const ExtTask*)tasks, 0, 8, o, w, stats, errf, 1u, 0u, n, opts);
    };
    WalkOut* wout2; u8* ops2;
    CHECK(hipMalloc(&wout2, (size_t)nmax * sizeof(WalkOut)));
    CHECK(hipMalloc(&ops2, (size_t)(nmax / 64) * MAXOPS * 64));
    // k_rcwalk3 against k_rcwalk2w on the same blocks: every WalkOut field, and (with kept ops) every op, must be equal
    auto same = [&](u32 n, int found) {
        ExtTask t[2]; memset(t, 0, sizeof t); t[0].found = found;
        CHECK(hipMemcpy(tasks, t, sizeof t, hipMemcpyHostToDevice));
        const u32 c2[4] = {n, 0, 0, 0};
        CHECK(hipMemcpy(ndev, c2, 16, hipMemcpyHostToDevice));
        CHECK(hipMemset(wout, 0xff, (size_t)n * sizeof(WalkOut))); CHECK(hipMemset(wout2, 0xee, (size_t)n * sizeof(WalkOut)));
        CHECK(hipMemset(ops, 0x7f, (size_t)(nmax / 64) * MAXOPS * 64)); CHECK(hipMemset(ops2, 0x7f, (size_t)(nmax / 64) * MAXOPS * 64));
        hipLaunchKernelGGL((k_rcwalk2w<NW, TW, N, MAXOPS>), dim3((n + 63) / 64), dim3(256), 0, 0, (const BlockItem*)items, n, (const u32*)ndev, n, (const u64*)frag, (const ulonglong2*)ck,
                           (const u64*)hc, (const BlockResult*)res, (const ExtTask*)tasks, 0, 8, ops, wout, stats, errf, 1u, 0u, n, 0u);
        launch3(n, 0, ops2, wout2, 0u);
        CHECK(hipDeviceSynchronize());
        std::vector<WalkOut> a(n), b(n);
        CHECK(hipMemcpy(a.data(), wout, (size_t)n * sizeof(WalkOut), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b.data(), wout2, (size_t)n * sizeof(WalkOut), hipMemcpyDeviceToHost));
        std::vector<u8> oa((size_t)((n + 63) / 64) * MAXOPS * 64), ob(oa.size());
        CHECK(hipMemcpy(oa.data(), ops, oa.size(), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ob.data(), ops2, ob.size(), hipMemcpyDeviceToHost));
        size_t bad_w = 0, bad_o = 0;
        for (u32 x = 0; x < n; ++x) if (memcmp(&a[x], &b[x], sizeof(WalkOut))) { if (!bad_w) printf("  first WalkOut difference at block %u: n %d / %d, nmat %d / %d, hit %d / %d, acnt %d / %d\n", x, a[x].n, b[x].n, a[x].nmat, b[x].nmat, a[x].hit, b[x].hit, a[x].acnt, b[x].acnt); ++bad_w; }
        for (size_t i = 0; i < oa.size(); ++i) if (oa[i] != ob[i]) ++bad_o;
        int he = 0; CHECK(hipMemcpy(&he, errf, 4, hipMemcpyDeviceToHost));
        printf("%s == k_rcwalk2w on %u blocks (%s): %zu WalkOut records differ, %zu op bytes differ, err %d  %s\n", kern == 4 ? "k_rcwalk3/16" : "k_rcwalk3/32", n, found ? "lean" : "ops kept", bad_w, bad_o, he, (bad_w || bad_o || he) ? "MISMATCH" : "ok");
        return !(bad_w || bad_o || he);
    };
    bool ok = same(nmax, 1); ok = same(nmax, 0) && ok; ok = same(1000, 0) && ok;
    auto run3 = [&](u32 n, int found, u32 opts, u32 lds, const char* what) {
        ExtTask t[2]; memset(t, 0, sizeof t); t[0].found = found;
        CHECK(hipMemcpy(tasks, t, sizeof t, hipMemcpyHostToDevice));
        const u32 c2[4] = {n, 0, 0, 0};
        CHECK(hipMemcpy(ndev, c2, 16, hipMemcpyHostToDevice));
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CHECK(hipEventRecord(e0, 0));
            launch3(n, lds, ops, wout, opts);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r) best = std::min(best, ms);
        }
        int he = 0; CHECK(hipMemcpy(&he, errf, 4, hipMemcpyDeviceToHost));
        WalkOut w0; CHECK(hipMemcpy(&w0, wout, sizeof w0, hipMemcpyDeviceToHost));
        printf("%-10s %7u blocks (%4u workgroups) | %-58s | %8.1f us | %.2f ns per block | err %d, block 0: n %d nmat %d\n", kern == 4 ? "k_rcwalk3/16" : "k_rcwalk3/32", n, (n + 63) / 64, what, best * 1e3, best * 1e6 / n, he, w0.n, w0.nmat);
    };
    for (kern = 4; kern >= 3; --kern) {
    if (kern == 3) { ok = same(nmax, 1) && ok; ok = same(nmax, 0) && ok; }
    run3(nmax, 1, 0, 0, "lean");
    run3(nmax, 0, 0, 0, "ops kept");
    run3(nmax, 1, 8, 0, "lean, s_setprio 3");
    run3(nmax, 1, 16, 0, "lean, only the walking wave at s_setprio 3");
    run3(nmax, 1, 0, 0, "lean again");
    run3(nmax, 1, 0, 2u << 10, "lean, + 2 KB of dynamic LDS per workgroup");
    run3(nmax, 1, 0, 6u << 10, "lean, + 6 KB");
    run3(nmax, 1, 0, 12u << 10, "lean, + 12 KB");
    run3(nmax, 1, 0, 20u << 10, "lean, + 20 KB");
    run3(nmax, 1, 0, 36u << 10, "lean, + 36 KB");
    run3(110592, 1, 0, 0, "lean, half the list (the bench's typical big round)");
    run3(81920, 1, 0, 0, "lean, 5 workgroups per CU's worth");
    run3(16384, 1, 0, 0, "lean, one workgroup per CU");
    run3(64, 1, 0, 0, "lean, ONE workgroup");
    }
    auto run = [&](u32 n, int found, u32 opts, u32 lds, const char* what) {
        ExtTask t[2]; memset(t, 0, sizeof t); t[0].found = found;
        CHECK(hipMemcpy(tasks, t, sizeof t, hipMemcpyHostToDevice));
        const u32 c2[4] = {n, 0, 0, 0};
        CHECK(hipMemcpy(ndev, c2, 16, hipMemcpyHostToDevice));
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_rcwalk2w<NW, TW, N, MAXOPS>), dim3((n + 63) / 64), dim3(256), lds, 0, (const BlockItem*)items, n, (const u32*)ndev, n, (const u64*)frag, (const ulonglong2*)ck,
                               (const u64*)hc, (const BlockResult*)res, (const ExtTask*)tasks, 0, 8, ops, wout, stats, errf, 1u, 0u, n, opts);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r) best = std::min(best, ms);
        }
        int he = 0; CHECK(hipMemcpy(&he, errf, 4, hipMemcpyDeviceToHost));
        WalkOut w0; CHECK(hipMemcpy(&w0, wout, sizeof w0, hipMemcpyDeviceToHost));
        printf("%7u blocks (%4u workgroups) | %-58s | %8.1f us | %.2f ns per block | err %d, block 0: n %d nmat %d\n", n, (n + 63) / 64, what, best * 1e3, best * 1e6 / n, he, w0.n, w0.nmat);
    };
    printf("---- k_rcwalk2w (round 4)\n");
    run(nmax, 1, 0, 0, "tasks past their first run of matches (lean walk)");
    run(nmax, 0, 0, 0, "tasks before it (ops kept)");
    run(nmax, 1, 8, 0, "lean, s_setprio 3");
    run(nmax, 1, 1, 0, "lean, next segment's inputs prefetched");
    run(nmax, 1, 16, 0, "lean, only the walking wave at s_setprio 3");
    run(nmax, 1, 0, 0, "lean again");
    run(nmax, 1, 0, 8u << 10, "lean, 4 workgroups per CU (8 KB of dynamic LDS)");
    run(nmax, 1, 0, 21u << 10, "lean, 3 workgroups per CU");
    run(nmax, 1, 0, 48u << 10, "lean, 2 workgroups per CU");
    run(81920, 1, 0, 0, "lean, one full round of workgroups (5 per CU)");
    run(16384, 1, 0, 0, "lean, one workgroup per CU");
    run(64, 1, 0, 0, "lean, ONE workgroup");
#ifdef NECAT_RC_TIMING
    run(nmax, 1, 2, 0, "lean, the walk of every segment done twice");
    run(nmax, 1, 4, 0, "lean, the recompute of every segment done twice");
    run(64, 1, 2, 0, "ONE workgroup, walk twice");
    run(64, 1, 4, 0, "ONE workgroup, recompute twice");
#endif
    return ok ? 0 : 2;
}
