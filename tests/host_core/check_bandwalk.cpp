// check_bandwalk.cpp - CPU replay of the diagonal-band walk (necat_amd/csrc/ext_bandwalk.h: band_piece, band_walk_col - the per-lane cores
// of k_rcwalk3) against walk_block (dp_core.h, the walk every other kernel and the oracle comparison rest on).
// Random blocks of every geometry the kernels see (512 x 512, ragged, list B up to 794, 2048-bp blocks, tiny ones), error rates 0 - 35 %, long
// insertions / deletions that push the walk out of its 32 diagonals (the redo path), every tail-match length incl. 0, ops kept or not:
// the full decision matrix is computed with advance_block_rec, walked with walk_block, and then walked the way the kernel does it -
// segment by segment, a column's 32-diagonal record cut out of the TWO words the quad recomputes (band_piece; everything outside those two
// words is poisoned), one band_walk_col per column - and every output (n, nmat, m, hit, acnt, qcnt, tcnt, mcnt, every op) must be equal.
//   g++ -O2 -std=c++17 -I necat_amd/csrc -o check_bandwalk tests/host_core/check_bandwalk.cpp && ./check_bandwalk [blocks] [seed]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "dp_core.h"
#include "ext_bandwalk.h"
using namespace necat;

struct FullMat {
    int nw; std::vector<u64> A, B;
    void rec(int c, int b, u64& a, u64& bb) const { a = A[(size_t)c * nw + b]; bb = B[(size_t)c * nw + b]; }
};
struct VecSink { std::vector<int>* v; bool on; bool storing() const { return on; } void put(int i, int op) { if ((int)v->size() <= i) v->resize(i + 1, -1); (*v)[i] = op; } };

// one block the way k_rcwalk3<.., BW> walks it: segment by segment, the pair of words band_word_lo<BW> names, every column's record cut out of the pair's
// decision planes (band_piece2, the fast form wherever the kernel uses it), everything the walk must not look at poisoned, one band_walk_col / band_walk_col3 per column
template <int BW, class Put, class Rng>
int kernel_walk(int blk, const FullMat& fm, int nw, BandWalk& w, int mlen, bool store, Put& put, Rng& rng, long long& redo, long long& myredo, long long& segs)
{
    constexpr int P0 = BW / 2;
    const u32 mask = BW == 32 ? 0xffffffffu : ((1u << BW) - 1u);
    bool out = false;
    int prev_seg = 1 << 30;
    while (!out) {
        const int seg = w.c >> 5, c0 = seg * 32, xin = w.c - c0, d0 = w.r - w.c;
        if (seg == prev_seg) { ++redo; ++myredo; } prev_seg = seg; ++segs;
        w.p = P0;
        u32 ra[32], rb_[32];
        const int wl = band_word_lo<BW>(w.r), wh = wl + 1 < nw ? wl + 1 : nw - 1;      // the kernel's pair of words (ext_rcwalk3.h)
        const bool general = (c0 + d0 - P0 - 64 * wl) < 0;                            // some column's record starts above the pair: zero fill
        for (int x = 0; x <= xin; ++x) {
            const int col = c0 + x, rb = col + d0 - P0, S = rb - 64 * wl;
            u64 pa0, pb0, pa1, pb1; fm.rec(col, wl, pa0, pb0); fm.rec(col, wh, pa1, pb1);
            if (wh == wl) { pa1 = rng(); pb1 = rng(); }                               // (the top word twice: its second copy is never looked at)
            if (!general && (S < 0 || S >= 96)) { fprintf(stderr, "block %d: S = %d outside the fast form's range\n", blk, S); return 1; }
            u32 a, b;
            if (general) { a = band_piece2<true>((u32)pa0, (u32)(pa0 >> 32), (u32)pa1, (u32)(pa1 >> 32), S); b = band_piece2<true>((u32)pb0, (u32)(pb0 >> 32), (u32)pb1, (u32)(pb1 >> 32), S); }
            else { a = band_piece2<false>((u32)pa0, (u32)(pa0 >> 32), (u32)pa1, (u32)(pa1 >> 32), S); b = band_piece2<false>((u32)pb0, (u32)(pb0 >> 32), (u32)pb1, (u32)(pb1 >> 32), S); }
            if (BW == 32 && (blk & 8) == 0) {      // and the four-lane form's pieces (band_piece per word, OR-ed): the same record where the walk can look
                u32 a2 = 0, b2 = 0;
                const int w1 = w.r >> 6;
                for (int k = 0; k < 2; ++k) {
                    const int ww = w1 - 1 + k;
                    if (ww < 0) continue;
                    u64 qa, qb; fm.rec(col, ww, qa, qb);
                    a2 |= band_piece(qa, rb - 64 * ww); b2 |= band_piece(qb, rb - 64 * ww);
                }
                for (int p = 0; p < 32; ++p) { const int row = rb + p; if (row <= w.r && (((a ^ a2) | (b ^ b2)) >> p & 1u)) { fprintf(stderr, "block %d: the two forms of the record differ at row %d\n", blk, row); return 1; } }
            }
            a &= mask; b &= mask;          // a record keeps BW diagonals (the kernel packs A | B << 16 for BW = 16)
            // rows above r (the walk only moves up): poison
            for (int p = 0; p < BW; ++p) { const int row = rb + p; if (row > w.r) { a = (a & ~(1u << p)) | ((u32)(rng() & 1) << p); b = (b & ~(1u << p)) | ((u32)(rng() & 1) << p); } }
            ra[x] = a; rb_[x] = b;
        }
        int st = 0;
        if ((blk & 6) == 6 || BW != 32) {    // the wave form: alive flag, the reason read off the state afterwards
            int ovf = 0; bool alive = true;
            for (int x = 31; x >= 0; --x) band_walk_col3<1 << 20, BW>(w, alive, x <= xin, x <= xin ? ra[x] : ((u32)rng() & mask), x <= xin ? rb_[x] : ((u32)rng() & mask), mlen, store, put, ovf);
            if (ovf) { fprintf(stderr, "block %d: op index overflow\n", blk); return 1; }
            st = band_walk_why(w, alive);
        } else
        for (int x = xin; x >= 0 && st == 0; --x) st = band_walk_col(w, ra[x], rb_[x], mlen, store, put);
        if (st == 2) out = true;
        else if (st == 0 && w.c < 0) { fprintf(stderr, "block %d: column ran out without the walk noticing\n", blk); return 1; }
    }
    return 0;
}

int main(int argc, char** argv)
{
    const int nblocks = argc > 1 ? atoi(argv[1]) : 4000;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], 0, 10) : 12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    long long redo = 0, segs = 0, maxredo = 0;
    for (int blk = 0; blk < nblocks; ++blk) {
        // ---- geometry
        int qn, tn;
        switch (blk % 6) {
        case 0: qn = tn = 512; break;
        case 1: qn = 1 + (int)(rng() % 512); tn = 1 + (int)(rng() % 512); break;
        case 2: qn = 513 + (int)(rng() % 282); tn = 400 + (int)(rng() % 395); break;
        case 3: qn = 1 + (int)(rng() % 80); tn = 1 + (int)(rng() % 80); break;
        case 4: qn = 2048; tn = 1900 + (int)(rng() % 300); break;
        default: qn = 448 + (int)(rng() % 65); tn = 448 + (int)(rng() % 65); break;
        }
        const double err = (blk % 7 == 0) ? 0.0 : (blk % 7 == 1) ? 0.35 : 0.02 + 0.2 * U(rng);
        const bool bursts = blk % 5 == 0;                      // long indels: the walk leaves its 32 diagonals
        std::vector<int> q(qn), t;
        for (auto& x : q) x = (int)(rng() & 3);
        {
            int qi = 0;
            while ((int)t.size() < tn) {
                if (qi >= qn) { t.push_back((int)(rng() & 3)); continue; }
                const double r = U(rng);
                if (bursts && r < 0.004) { const int k = 10 + (int)(rng() % 60); if (rng() & 1) qi += k; else for (int i = 0; i < k; ++i) t.push_back((int)(rng() & 3)); continue; }
                if (r < err / 3) { t.push_back((q[qi] + 1 + (int)(rng() % 3)) & 3); ++qi; }
                else if (r < 2 * err / 3) t.push_back((int)(rng() & 3));
                else if (r < err) ++qi;
                else t.push_back(q[qi++]);
            }
            t.resize(tn);
        }
        // ---- the full decision matrix (every word of every column; SHW: the top row's boundary is +1 per column)
        const int nw = (qn + 63) / 64, W = nw * 64 - qn;
        FullMat fm; fm.nw = nw; fm.A.assign((size_t)tn * nw, 0); fm.B.assign((size_t)tn * nw, 0);
        {
            std::vector<u64> P(nw, ~0ULL), M(nw, 0ULL);
            for (int c = 0; c < tn; ++c) {
                int h = 1;
                for (int b = 0; b < nw; ++b) {
                    u64 eq = 0;
                    for (int i = 0; i < 64; ++i) { const int r = 64 * b + i; if (r < qn ? q[r] == t[c] : true) eq |= 1ULL << i; }
                    (void)W;
                    h = advance_block_rec(P[b], M[b], eq, h, P[b], M[b], fm.A[(size_t)c * nw + b], fm.B[(size_t)c * nw + b]);
                }
            }
        }
        const int endc = (blk % 3 == 0) ? tn - 1 : (int)(rng() % tn);      // any end column: the decisions are exact everywhere
        static const int mlens[5] = {8, 1, 4, 0, 13};
        const int mlen = mlens[blk % 5];
        const bool store = (blk & 1) != 0;
        // ---- reference: walk_block from (qn - 1, endc)
        std::vector<int> ops_ref, ops_new;
        TailScan ts; tail_init(ts, mlen);
        { VecSink sk{&ops_ref, store}; walk_block(qn, endc + 1, fm, sk, ts); }
        // ---- the kernel's way, on records of 32 diagonals (k_rcwalk3<.., 32>) or of 16 (k_rcwalk3<.., 16>: half the LDS per block in flight)
        BandWalk w; memset(&w, 0, sizeof w); w.r = qn - 1; w.c = endc;
        VecSink sk{&ops_new, store};
        auto put = [&](int i, int op) { sk.put(i, op); };
        long long myredo = 0;
        const int bw = (blk & 16) ? 16 : 32;
        const int rcw = bw == 16 ? kernel_walk<16>(blk, fm, nw, w, mlen, store, put, rng, redo, myredo, segs) : kernel_walk<32>(blk, fm, nw, w, mlen, store, put, rng, redo, myredo, segs);
        if (rcw) return rcw;
        if (myredo > maxredo) maxredo = myredo;
        {   // the walk's epilogue (k_rcwalk2w / walk_block): the rest of the other sequence is one run of inserts / deletes
            const int kop = w.c < 0 ? 1 : 2, k = w.c < 0 ? w.r + 1 : w.c + 1;
            if (store) for (int i = 0; i < k; ++i) put(w.n + i, kop);
            w.n += k;
            if (!w.hit && k > 0) w.m = 0;
        }
        const bool same = w.n == ts.n && w.nmat == ts.nmat && w.m == ts.m && w.hit == ts.hit && w.acnt == ts.acnt && w.qcnt == ts.qcnt && w.tcnt == ts.tcnt && w.mcnt == ts.mcnt && ops_new == ops_ref;
        if (!same) {
            fprintf(stderr, "block %d (%d x %d, end %d, mlen %d, store %d): n %d/%d nmat %d/%d m %d/%d hit %d/%d acnt %d/%d qcnt %d/%d tcnt %d/%d mcnt %d/%d ops %s\n", blk, qn, tn, endc, mlen,
                    (int)store, w.n, ts.n, w.nmat, ts.nmat, w.m, ts.m, w.hit, ts.hit, w.acnt, ts.acnt, w.qcnt, ts.qcnt, w.tcnt, ts.tcnt, w.mcnt, ts.mcnt, ops_new == ops_ref ? "equal" : "DIFFER");
            return 1;
        }
    }
    printf("check_bandwalk: %d blocks equal to walk_block; %lld segment passes, %lld of them redone (at most %lld in one block)\n", nblocks, segs, redo, maxredo);
    return 0;
}
