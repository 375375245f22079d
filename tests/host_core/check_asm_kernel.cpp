// check_asm_kernel.cpp - TEST ONLY.  The SOURCE of the device's block aligner for oc2asmpm (necat_amd/csrc/asm_kernels.h: k_asm_align, one lane = one
// alignment) compiled with g++ behind a few lines that stand in for the HIP built-ins, and run lane by lane, wave by wave, exactly as the library launches
// it (necat_asm_align_batch: shared band / op slabs per wave, per-anchor column regions, the alignment put together from the two streams) - against the
// oracle's onc_align at block size 2048 / tail match length 8.  Catches kernel-logic regressions on a machine without a GPU.
//
//   check_asm_kernel <n pairs> <seed>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <string>
#include <vector>

// ---- what the kernel source needs from HIP
struct ulonglong2 { unsigned long long x, y; };
static inline ulonglong2 make_ulonglong2(unsigned long long x, unsigned long long y) { ulonglong2 v; v.x = x; v.y = y; return v; }
struct FakeIdx { unsigned x = 0, y = 0, z = 0; };
static FakeIdx blockIdx, threadIdx;
#define __global__
#define __launch_bounds__(...)

#define NECAT_XCHECK 1          // k_asm_align is a kernel of the cross-check build (necat_hip.hip, NECAT_BUILD_CROSSCHECK)
#include "../../necat_amd/csrc/asm_kernels.h"
extern "C" {
#include "../../oracle/necat_oracle.h"
}
using namespace necat;

static std::vector<uint8_t> mutate(const std::vector<uint8_t>& g, double err, std::mt19937_64& r)
{
    std::vector<uint8_t> q;
    std::uniform_real_distribution<double> u(0, 1);
    for (uint8_t c : g) {
        const double x = u(r);
        if (x < err / 3) q.push_back((uint8_t)((c + 1 + r() % 3) & 3));
        else if (x < 2 * err / 3) { q.push_back(c); q.push_back((uint8_t)(r() & 3)); }
        else if (x < err) continue;
        else q.push_back(c);
    }
    return q;
}

int main(int argc, char** argv)
{
    const int npairs = argc > 1 ? atoi(argv[1]) : 40;
    std::mt19937_64 rng(argc > 2 ? (uint64_t)atoll(argv[2]) : 7);
    // one volume holding every sequence's FORWARD strand; the alignment sees the subject on strand sdir
    struct Row { int q, s, sdir, qoff, soff; std::vector<uint8_t> qs, ts; };
    std::vector<std::vector<uint8_t>> seqs;
    std::vector<Row> rows;
    for (int it = 0; it < npairs; ++it) {
        std::vector<uint8_t> g((size_t)(1500 + rng() % 9000));
        for (auto& c : g) c = (uint8_t)(rng() & 3);
        const double e = it % 4 ? 0.005 + 0.05 * (double)(rng() % 1000) / 1000.0 : 0.10 + 0.06 * (double)(rng() % 1000) / 1000.0;
        std::vector<uint8_t> q = mutate(g, e, rng), t = mutate(g, e, rng);
        if (it % 9 == 4) for (auto& c : t) c = (uint8_t)(rng() & 3);
        const int sdir = it & 1;
        std::vector<uint8_t> stored = t;
        if (sdir) { std::reverse(stored.begin(), stored.end()); for (auto& c : stored) c = (uint8_t)(3 - c); }
        const int qid = (int)seqs.size(), sid = qid + 1;
        seqs.push_back(q); seqs.push_back(stored);
        for (int k = 0; k < 3; ++k) {
            double frac = it % 5 ? (double)(rng() % 100000) / 100000.0 : (double)(rng() & 1);
            rows.push_back(Row{qid, sid, sdir, (int)(frac * (double)(q.size() - 1)), (int)(frac * (double)(t.size() - 1)), q, t});
        }
    }
    const int G = 4;     // guard words, as the library's volumes have them (runtime.h: kGuardWords)
    std::vector<u64> off(seqs.size() + 1, 0);
    for (size_t i = 0; i < seqs.size(); ++i) off[i + 1] = off[i] + seqs[i].size();
    std::vector<u64> words((off.back() + 31) / 32 + 2 * G, 0);
    for (size_t i = 0; i < seqs.size(); ++i) for (size_t k = 0; k < seqs[i].size(); ++k) { const u64 g = off[i] + k; words[G + (g >> 5)] |= (u64)seqs[i][k] << ((g & 31) * 2); }
    DevVolume vol; vol.bases = words.data() + G; vol.seq_off = off.data(); vol.nbases = off.back(); vol.nseq = seqs.size();
    // the launch of necat_asm_align_batch
    const size_t n = rows.size();
    std::vector<AsmAnchor> anchors(n);
    std::vector<u64> coff(n + 1, 0);
    for (size_t i = 0; i < n; ++i) {
        anchors[i] = AsmAnchor{rows[i].q, rows[i].s, rows[i].sdir, rows[i].qoff, rows[i].soff};
        coff[i + 1] = coff[i] + ((rows[i].qs.size() + rows[i].ts.size() + 64 + 7) & ~7ULL);
    }
    std::vector<u8> cols(coff[n] + 8, 0xff);
    std::vector<AsmOut> out(n);
    const u32 waves_total = (u32)((n + 63) / 64), waves_max = 2;       // two waves per "launch": the slabs are reused by the next one
    std::vector<char> band((size_t)waves_max * kAsmBandWave);
    std::vector<u8> opsp((size_t)waves_max * kAsmOpsWave);
    for (u32 w0 = 0; w0 < waves_total; w0 += waves_max) {
        const u32 nw = std::min(waves_max, waves_total - w0);
        const u64 first = (u64)w0 * 64, cnt = std::min<u64>((u64)nw * 64, n - first);
        for (u32 w = 0; w < nw; ++w) for (unsigned lane = 0; lane < 64; ++lane) {
            blockIdx.x = w; threadIdx.x = lane;
            k_asm_align(anchors.data() + first, (u32)cnt, vol, vol, 0.5, 8, band.data(), opsp.data(), cols.data(), coff.data() + first, out.data() + first);
        }
    }
    // against the oracle
    ora_aligner* al = ora_aligner_new(0.5);
    long bad = 0, n_ok = 0, n_empty = 0, n_blocks = 0;
    for (size_t i = 0; i < n; ++i) {
        const Row& r = rows[i];
        ora_align_result ar;
        const int ok = ora_onc_align(al, r.qs.data(), r.qoff, (int)r.qs.size(), r.ts.data(), r.soff, (int)r.ts.size(), 2048, 400, 8, &ar);
        const AsmOut& o = out[i];
        const int nl = o.lto - o.lfrom, nr = o.rto - o.rfrom;
        bool same = o.err == 0 && nl >= 0 && nr >= 0 && nl + nr == o.cols && o.cols == ar.align_size && (o.cols >= 400) == (ok != 0) &&
                    (o.cols == 0 || (o.qoff == ar.qoff && o.qend == ar.qend && o.toff == ar.toff && o.tend == ar.tend));
        if (same && o.cols) same = 100.0 * (double)o.mat / (double)o.cols == ar.ident_perc;
        const u8* c = cols.data() + coff[i];
        for (int j = 0; same && j < nl + nr; ++j) {
            const u8 op = j < nl ? c[o.lto - 1 - j] : c[o.lto + o.rfrom + (j - nl)];
            const char qa = ar.query_align[j], ta = ar.target_align[j];
            const int want = qa == '-' ? 2 : (ta == '-' ? 1 : (qa == ta ? 0 : 3));
            same = op == want;
        }
        if (!same) { if (bad < 5) fprintf(stderr, "MISMATCH anchor %zu: kernel (%d %d %d %d cols %d err %d) oracle (%d %d %d %d cols %d)\n", i, o.qoff, o.qend, o.toff, o.tend, o.cols, o.err,
                                          ar.qoff, ar.qend, ar.toff, ar.tend, ar.align_size); ++bad; }
        n_ok += ok; n_empty += ar.align_size == 0; n_blocks += o.blocks;
    }
    printf("check_asm_kernel: anchors=%zu aligned=%ld empty=%ld blocks=%ld mismatches=%ld\n", n, n_ok, n_empty, n_blocks, bad);
    return bad ? 1 : 0;
}
