// check_chain.cpp - the wave-parallel chain DP of seed_kernels.h (chain_fill_wave), transcribed lane by lane for the host
// (the wave primitives become loops over 64-element arrays), against the sequential chain_fill of seed_core.h
// (= chain_dp.c:46-85) on random seed sets.  Test infrastructure; the GPU parity tests run the real kernel.
#include "../../necat_amd/csrc/seed_core.h"
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <vector>
#include <algorithm>
#include <random>
using namespace necat;

static void fill_lanes(const u64* cs, i32* f, i32* p, i32* t, i32* v, int n, int kmer_size)
{
    for (int a = 0; a < n; ++a) { f[a] = 0; p[a] = -1; t[a] = 0; v[a] = 0; }
    int st = 0;
    for (int i = 0; i < n; ++i) {
        const u64 ci = cs[i];
        const i64 ri = (i64)(ci >> 32);
        while (st < i && ri - (i64)(cs[st] >> 32) > kChainMaxDist) ++st;
        int max_f = kmer_size, max_j = -1, n_skip = 0;
        for (int top = i - 1; top >= st; top -= 64) {
            int sc[64]; bool valid[64], marked[64], newmax[64];
            for (int l = 0; l < 64; ++l) {          // every lane stores its mark first
                const int j = top - l;
                sc[l] = INT_MIN; valid[l] = false;
                if (j >= st) {
                    valid[l] = chain_pair_score(ci, cs[j], kmer_size, f[j], &sc[l]);
                    if (valid[l]) { if (p[j] >= 0) t[p[j]] = i; } else sc[l] = INT_MIN;
                }
            }
            for (int l = 0; l < 64; ++l) marked[l] = valid[l] && t[top - l] == i;      // after the barrier
            int incl[64];
            for (int l = 0; l < 64; ++l) incl[l] = l ? std::max(incl[l - 1], sc[l]) : sc[l];
            u64 NM = 0, SK = 0;
            for (int l = 0; l < 64; ++l) {
                int before = l ? incl[l - 1] : max_f;
                if (l == 0 || before < max_f) before = max_f;
                newmax[l] = valid[l] && sc[l] > before;
                if (newmax[l]) NM |= 1ULL << l;
                if (marked[l] && !newmax[l]) SK |= 1ULL << l;
            }
            u64 live = ~0ULL;
            if (SK) {
                int S[64], W[64]; u64 stop = 0;
                int mn = INT_MAX;
                for (int l = 0; l < 64; ++l) {
                    const u64 upto = (l == 63) ? ~0ULL : ((1ULL << (l + 1)) - 1);
                    S[l] = popc64(SK & upto) - popc64(NM & upto);
                    mn = std::min(mn, S[l]);
                    int fl = mn; if (fl > -n_skip) fl = -n_skip;
                    W[l] = S[l] - fl;
                    if (W[l] > kChainMaxSkip) stop |= 1ULL << l;
                }
                if (stop) live = (1ULL << ctz64(stop)) - 1ULL;
                else n_skip = W[63];
            } else {
                n_skip -= popc64(NM); if (n_skip < 0) n_skip = 0;
            }
            const u64 best = NM & live;
            if (best) { const int lb = 63 - __builtin_clzll(best); max_f = sc[lb]; max_j = top - lb; }
            if (live != ~0ULL) break;
        }
        f[i] = max_f; p[i] = max_j;
        v[i] = (max_j >= 0 && v[max_j] > max_f) ? v[max_j] : max_f;
    }
}

int main(int argc, char** argv)
{
    const int trials = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937_64 rng(12345);
    long bad = 0, stops = 0, total = 0;
    for (int tr = 0; tr < trials; ++tr) {
        const int n = 2 + (int)(rng() % (tr % 7 == 0 ? 900 : 260));
        const int kmer = (tr % 3 == 0) ? 15 : 13;
        const int mode = tr % 5;
        std::vector<u64> cs(n);
        // seeds near a few diagonals, dense (many candidates per scan and long runs of already-marked predecessors)
        for (int a = 0; a < n; ++a) {
            const i64 diag = (i64)(rng() % (mode == 0 ? 3 : 40)) * (mode == 1 ? 7 : 120);
            const i64 q = (i64)(rng() % (mode == 2 ? 800 : 6000));
            i64 s = q + diag + (i64)(rng() % 9) - 4;
            if (s < 0) s = 0;
            cs[a] = ((u64)s << 32) | (u64)(u32)q;
        }
        std::sort(cs.begin(), cs.end());
        std::vector<i32> f1(n), p1(n), t1(n), v1(n), f2(n), p2(n), t2(n), v2(n);
        SeedScratch S; S.cs = cs.data(); S.f = f1.data(); S.p = p1.data(); S.t = t1.data(); S.v = v1.data();
        chain_fill(S, n, kmer);
        fill_lanes(cs.data(), f2.data(), p2.data(), t2.data(), v2.data(), n, kmer);
        for (int a = 0; a < n; ++a) { ++total; if (f1[a] != f2[a] || p1[a] != p2[a] || v1[a] != v2[a]) { if (bad < 5) fprintf(stderr, "trial %d seed %d: f %d/%d p %d/%d v %d/%d\n", tr, a, f1[a], f2[a], p1[a], p2[a], v1[a], v2[a]); ++bad; } }
        // how often the max_skip stop fires in the sequential loop (the test must exercise it)
        {
            std::vector<i32> f(n, 0), p(n, -1), t(n, 0);
            int st = 0;
            for (int i = 0; i < n; ++i) {
                while (st < i && (i64)(cs[i] >> 32) - (i64)(cs[st] >> 32) > kChainMaxDist) ++st;
                int max_f = kmer, max_j = -1, n_skip = 0;
                for (int j = i - 1; j >= st; --j) {
                    int sc; if (!chain_pair_score(cs[i], cs[j], kmer, f[j], &sc)) continue;
                    if (sc > max_f) { max_f = sc; max_j = j; if (n_skip > 0) --n_skip; }
                    else if (t[j] == i) { if (++n_skip > kChainMaxSkip) { ++stops; break; } }
                    if (p[j] >= 0) t[p[j]] = i;
                }
                f[i] = max_f; p[i] = max_j;
            }
        }
    }
    printf("seeds %ld mismatches %ld max_skip stops %ld\n", total, bad, stops);
    return bad ? 1 : (stops ? 0 : 3);
}
