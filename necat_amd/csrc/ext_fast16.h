// ext_fast16.h - list-A DP kernel for waves of SIXTEEN full blocks (512 x 512): the SHW pass as in myers_fast_full (8 lanes
// per block, two halves of 8 blocks one after the other), the NW pass with FOUR lanes per block on a sliding window of words.
//
// Why: the NW pass only has to produce the band the walk can stand on.  With k = the block's distance and d = 512 - (end
// column + 1), a cell (r, c) can lie on an alignment of cost <= k only if |r - c| + |d - (r - c)| <= k (Ukkonen): a diagonal
// band of k + 1 rows, i.e. 2 - 3 of the 8 words per column (the reference's own band: 1.9 words) - but with one
// lane per word the other 5 lanes of the block computed (and discarded) their words all the same: 4 x the needed work in
// the more expensive of the two passes.  Here word w is live only for the columns c with
//     64 w - hi <= c <= 64 w + 63 - lo,      [lo, hi] = the band's range of r - c,
// at most 64 + k + 1 <= 245 columns (blocks with k > 180 take the 8-lane NW pass of myers_fast_full afterwards), so words w
// and w + 4 are never live together and four lanes - lane j takes word j, then word j + 4 - cover the band.  The
// anti-diagonal schedule is unchanged (word w computes column c at step c + w), the carry into word w comes from the lane of
// word w - 1 (a rotation inside the quad: DPP quad_perm), a word whose upper neighbour is not live gets the boundary carry
// (+1), and a word entering the band starts from P = all ones / M = 0 - the band growth rule of the reference (edlib_ex.c:
// 303-318).  Every live (word, column) is stored: the walk reads exactly the same decisions as before, because its every cell -
// and every neighbour it could move to - lies on an optimal alignment, hence inside the band, hence is exact in any band that
// contains Ukkonen's (dp_core.h, traceback notes).
#pragma once

namespace necat {

constexpr int kNarrowMaxDist = 180;      // blocks up to this distance go through the 4-lane NW pass

NECAT_D u32 dpp_quad_rot1(u32 v)         // lane i of every quad receives v of lane (i - 1) & 3
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x93 /* quad_perm:[3,0,1,2] */, 0xf, 0xf, false);
}

// SHW pass of 8 blocks (lane = 8 sub + b), see myers_fast_full.  Returns min over columns of (bottom-row value << 10 | step) of
// the lane's word; the block's result is word 7's.
template <int TW>
NECAT_D u32 fast_shw8(const int b, const u64* __restrict__ tw, const u64 nlo, const u64 nhi)
{
    constexpr int G = 8, N = kOcaBlockSize, kSteps = N + G - 1;
    const u32 cm = b == G - 1 ? 0x80000000u : 0u;
    const u32 nlo_l = (u32)nlo, nlo_h = (u32)(nlo >> 32), nhi_l = (u32)nhi, nhi_h = (u32)(nhi >> 32);
    const u32 sk = (u32)(32 - b) & 31u;
    u32 tlo = 0, thi = 0, plo = 0, phi = 0;
    FastWord w; w.Pv = ~0ULL; w.Mv = 0ULL; w.pubP = 0x80000000u; w.pubM = 0u;
    u32 S = (u32)(b + 1) * 64u, key = 0xffffffffu;
    u64 dA, dB;
    u32 cph = 0x80000000u, cmh = 0u;
    for (int s0 = 0; s0 < kSteps; s0 += 32) {
        {
            const u64 x = (s0 >> 5) < TW ? tw[s0 >> 5] : 0ULL;
            const u32 xl = (u32)x, xh = (u32)(x >> 32);
            tlo = b ? __builtin_amdgcn_alignbit(xl, plo, sk) : xl;
            thi = b ? __builtin_amdgcn_alignbit(xh, phi, sk) : xh;
            plo = xl; phi = xh;
        }
        const int jn = kSteps - s0 < 32 ? kSteps - s0 : 32;
        for (int j = 0; j < jn; ++j) {
            const int s = s0 + j;
            cph = dpp_row_shr1(w.pubP, cph); cmh = dpp_row_shr1(w.pubM, cmh);
            const bool edge = s < G - 1 || s >= N;
            if (!edge || (s >= b && s - b < N)) {
                const u32 ma = (u32)__builtin_amdgcn_sbfe((int)tlo, (u32)j, 1u), mb = (u32)__builtin_amdgcn_sbfe((int)thi, (u32)j, 1u);
                const u32 el = bop<0x60>(nlo_l ^ ma, nhi_l, mb), eh = bop<0x60>(nlo_h ^ ma, nhi_h, mb);
                u32 phh, mhh;
                fast_advance<false>(w, el, eh, cph, cmh, cm, phh, mhh, dA, dB);
                S += (phh >> 31) - (mhh >> 31);
                const u32 k2 = (S << 10) + (u32)s;
                key = k2 < key ? k2 : key;
            }
        }
    }
    return key;
}

// one workgroup (two waves) = 16 consecutive items of one slab, all of them full blocks; lane = lane inside the wave.
//   fr0   : fragment words of item 0 of the wave (lane stride 1 between items, word stride 64)
//   slab  : band records of the 64-item group;  il0 = index of the wave's first item inside that group (a multiple of 16)
//   tl    : LDS, target bit-planes of the 16 blocks [16][TW];  res : LDS, (best, end0) per block
template <int NW, int TW>
NECAT_D void myers_fast16(const int lane, const u64* __restrict__ fr0, char* __restrict__ slab, const int il0, const double error,
                          BlockResult* __restrict__ results, unsigned long long* __restrict__ stats, u64 (*tl)[TW], int (*res)[2], const bool no_store)
{
    constexpr int N = kOcaBlockSize;
    // ---- target planes of the 16 blocks -> LDS (both waves of the workgroup)
    for (int e = (int)threadIdx.x; e < 16 * TW; e += 128) {
        const int q = e / TW, w = e % TW;
        const u64 x = fr0[(u64)(2 * NW + w) * 64 + q];
        tl[q][w] = even_bits(x) | (even_bits(x >> 1) << 32);
    }
    __syncthreads();
    // ---- SHW, 8 lanes per block: wave h takes blocks 8 h .. 8 h + 7, the two waves side by side
    const int k0 = (int)((double)N * error * 1.1);                       // edlib_ex.c:751
    {
        const int h = (int)(threadIdx.x >> 6);
        const int sub = lane >> 3, b = lane & 7, q = 8 * h + sub;
        const u64 nlo = fr0[(u64)b * 64 + q], nhi = fr0[(u64)(NW + b) * 64 + q];
        const u32 key = fast_shw8<TW>(b, tl[q], nlo, nhi);
        if (b == 7) {
            int best = (int)(key >> 10);
            const int end0 = (int)(key & 1023u) - 7;
            if (best > k0) best = -1;
            res[q][0] = best; res[q][1] = end0;
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;               // the NW pass of all 16 blocks fits one wave (4 lanes per block): the second wave is done
    // ---- NW, 4 lanes per block
    const int q = lane >> 2, j = lane & 3;
    const int best = res[q][0], end0 = res[q][1];
    const int tn2 = end0 + 1, d = N - tn2;
    int err = 0;
    if (best >= 0) { const int ad = d < 0 ? -d : d; if (best < ad) err = 1; }
    const bool go = best >= 0 && !err;
    const bool narrow = go && best <= kNarrowMaxDist;
    const int ad = d < 0 ? -d : d;
    const int slack = go ? (best - ad) >> 1 : 0;
    const int lo_x = (d < 0 ? d : 0) - slack, hi_x = (d > 0 ? d : 0) + slack;          // the band: lo_x <= r - c <= hi_x
    int steps = narrow ? tn2 + NW - 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(steps, o); steps = x > steps ? x : steps; }
    steps = __builtin_amdgcn_readfirstlane(steps);
    // this lane's two words: j, then j + 4
    const u64 nloA = fr0[(u64)j * 64 + q], nhiA = fr0[(u64)(NW + j) * 64 + q];
    const u64 nloB = fr0[(u64)(j + 4) * 64 + q], nhiB = fr0[(u64)(NW + j + 4) * 64 + q];
    u32 nlo_l = (u32)nloA, nlo_h = (u32)(nloA >> 32), nhi_l = (u32)nhiA, nhi_h = (u32)(nhiA >> 32);
    int w = j;                                                           // current word
    auto first_col = [&](int ww) { const int c = 64 * ww - hi_x; return c < 0 ? 0 : c; };
    auto last_col = [&](int ww) { const int c = 64 * ww + 63 - lo_x; return c > tn2 - 1 ? tn2 - 1 : c; };
    int c0 = first_col(w), c1 = narrow ? last_col(w) : -1;               // live columns of the current word
    u32 nlive = c1 >= c0 ? (u32)(c1 - c0 + 1) : 0u;                      // how many (0: never live)
    const int il = il0 + q;
    char* const rbase = slab;
    auto rec_off = [&](int c, int ww) -> u32 { return (((((u32)c >> 3) * (u32)NW + (u32)ww) * 64u + (u32)il) * 8u + ((u32)c & 7u)) * 16u; };
    // top word of the band: no live word above -> boundary carry.  Word 0 never has one; the others lose theirs when the word above
    // retires, which the lane above signals by publishing the boundary carry itself.
    // (the lane above publishes the boundary carry while it is between its two words; once it has started word w + 3 this lane
    // must already have switched its own masks - checked once per group of 8 steps, the gap is >= 16 steps)
    u32 or_p = w == 0 ? 0x80000000u : 0u, and_m = w == 0 ? 0u : ~0u;
    int c_up = w > 0 ? last_col(w - 1) : -1;                            // last live column of the word above
    FastWord wd; wd.Pv = ~0ULL; wd.Mv = 0ULL; wd.pubP = 0x80000000u; wd.pubM = 0u;
    u32 xl = 0, xh = 0, pl = 0, ph = 0, tlo = 0, thi = 0;                // raw planes of the current / previous 32-column window, skewed window
    auto skew = [&]() {
        const u32 sk = (u32)(32 - w) & 31u;
        tlo = w ? __builtin_amdgcn_alignbit(xl, pl, sk) : xl;
        thi = w ? __builtin_amdgcn_alignbit(xh, ph, sk) : xh;
    };
    u32 kept = 0;
    u64 rA, rB;
    for (int s0 = 0; s0 < steps; s0 += 8) {
        if ((s0 & 31) == 0) {
            const u64 x = (s0 >> 5) < TW ? tl[q][s0 >> 5] : 0ULL;
            pl = xl; ph = xh; xl = (u32)x; xh = (u32)(x >> 32);
            skew();
        }
        // retire a finished word (checked once per group of 8 steps: the next word of the lane starts >= 16 steps later)
        if (w < 4 && s0 - w > c1) {
            w += 4;
            nlo_l = (u32)nloB; nlo_h = (u32)(nloB >> 32); nhi_l = (u32)nhiB; nhi_h = (u32)(nhiB >> 32);
            c0 = first_col(w); c1 = narrow ? last_col(w) : -1;
            nlive = c1 >= c0 ? (u32)(c1 - c0 + 1) : 0u;
            wd.Pv = ~0ULL; wd.Mv = 0ULL;
            or_p = 0u; and_m = ~0u;
            c_up = last_col(w - 1);
            skew();
        }
        if (s0 - w > c_up) { or_p = 0x80000000u; and_m = 0u; }           // the word above has retired: boundary carry from here on
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int s = s0 + t;
            const u32 cph = dpp_quad_rot1(wd.pubP) | or_p, cmh = dpp_quad_rot1(wd.pubM) & and_m;
            wd.pubP = 0x80000000u; wd.pubM = 0u;                         // a lane that is not live publishes the boundary carry
            const int c = s - w;
            if ((u32)(c - c0) < nlive) {
                const u32 jj = (u32)s & 31u;
                const u32 ma = (u32)__builtin_amdgcn_sbfe((int)tlo, jj, 1u), mb = (u32)__builtin_amdgcn_sbfe((int)thi, jj, 1u);
                const u32 el = bop<0x60>(nlo_l ^ ma, nhi_l, mb), eh = bop<0x60>(nlo_h ^ ma, nhi_h, mb);
                u32 phh, mhh;
                fast_advance<true>(wd, el, eh, cph, cmh, 0u, phh, mhh, rA, rB);
                if (!no_store) *reinterpret_cast<ulonglong2*>(rbase + rec_off(c, w)) = make_ulonglong2(rA, rB);
                ++kept;
            }
        }
    }
    // ---- results (one lane per block) and work counters
    u32 live = kept;
    for (int o = 32; o > 0; o >>= 1) live += (u32)__shfl_xor((int)live, o);
    if (lane == 0) { stat_add(stats, 2, (unsigned long long)live); stat_add(stats, 0, (unsigned long long)live + 16ull * NW * N); stat_add(stats, 1, 16ull * 2 * N); }
    if (j == 0 && (narrow || !go)) {
        BlockResult br; br.dist = go ? best : -1; br.endc = end0; br.err = err;
        br.words = (u32)(NW * N) + (narrow ? (u32)((64 + hi_x - lo_x) * NW) : 0u);
        results[q] = br;
    }
    // ---- the rare wide blocks (distance > kNarrowMaxDist): the 8-lane NW pass of myers_fast_full, half by half
    const bool wide = go && !narrow;
    if (__any(wide)) {
        for (int h = 0; h < 2; ++h) {
            const int sub = lane >> 3, b = lane & 7, qq = 8 * h + sub;
            const int bestw = res[qq][0], end0w = res[qq][1];
            int errw = 0;
            if (bestw >= 0) { int a2 = N - (end0w + 1); if (a2 < 0) a2 = -a2; if (bestw < a2) errw = 1; }
            const bool gow = bestw > kNarrowMaxDist && !errw;
            if (!__any(gow)) continue;
            const u64 nlo = fr0[(u64)b * 64 + qq], nhi = fr0[(u64)(NW + b) * 64 + qq];
            fast_nw8<NW, TW>(lane, tl[qq], nlo, nhi, reinterpret_cast<ulonglong2*>(slab), il0 + qq, gow ? bestw : -1, end0w, gow, stats, no_store);
            if (b == 7 && gow) {
                BlockResult br; br.dist = bestw; br.endc = end0w; br.err = 0; br.words = (u32)(NW * (N + end0w + 1));
                results[qq] = br;
            }
        }
    }
}

}  // namespace necat
