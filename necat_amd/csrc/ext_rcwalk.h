// ext_rcwalk.h - full 512 x 512 blocks WITHOUT the NW pass, the band in HBM and the lane-per-block walk: the walk recomputes the
// cells it stands on.
//
// What the three-kernel chain spends on a full block (E. coli bench, a round of 220 k blocks): SHW pass 0.95 ms, NW pass 0.95 ms
// (all 8 words of every column recomputed to store ~ 1.9 of them: 1.9 GB of band records), walk 1.1 - 1.26 ms (one dependent
// 16-byte load per step from those records, 2.2 us per step with 228 k walks in flight).  Here:
//   * k_myers_ck        the SHW pass of the fast path (myers_fast_full) that also keeps CHECKPOINTS: every word's (Pv, Mv) after columns
//                       31, 63, .., 479 - 16 bytes per lane every 32 steps, 2 KB per block, 0.46 GB per round instead of 1.9 GB of records;
//   * k_rcwalk4         4 lanes per block, 16 blocks per wave, from the end cell backwards one 32-column SEGMENT at a time: the segment's
//                       columns are recomputed forward from the checkpoint before it on a window of 4 words that contains the block's
//                       Ukkonen band there (cells an alignment of cost <= the block's distance can pass, |r - c| + |d - (r - c)| <= best:
//                       the same window argument as the 4-lane NW pass of ext_fast16.h - exact where the walk looks, band boundary
//                       carry (+1) at the window's top), the walk's decisions for the 64 rows at and above the walker are cut out of
//                       the records into LDS (16 bytes per column), and the walk runs on them until it leaves the segment (or, rarely,
//                       those 64 rows: the segment is then redone from the walker's new row);
//   * k_traceback<WALK = 5>   the walk's statistics (TailScan) and ops taken from k_rcwalk4 instead of walked: trimming, counters, kept
//                       columns, next block - unchanged code.
// Word updates per block: 4096 (SHW) + ~ 2 200 (16 segments x 35 steps x 4 lanes) instead of 8192; no band records; every
// memory access of the walk is an LDS read.  Blocks whose distance is too large for the 4-word window (best > kRcMaxDist: < 0.1 % at
// 12 % error) are flagged by k_myers_ck and go through the old kernels (`only wide` launches of k_myers_coop / k_traceback).
#pragma once

namespace necat {

constexpr int kRcSeg = 32;                       // columns per segment = per checkpoint
constexpr int kRcCk = kOcaBlockSize / kRcSeg;    // checkpoint slots per block (the last one is never read)
constexpr int kRcCk16 = kOcaBlockSize / 16;      // .. of the CARRY variant
constexpr int kRcMaxDist = 160;                  // 63 (alignment of the window) + 31 (columns) + best <= 255 rows of a 4-word window

// ---- SHW with checkpoints (fast_shw8 of ext_fast16.h + one 16-byte store per lane every 32 steps)
// CARRY (the 2-lane-window walk k_rcwalk2 below): checkpoints every 16 columns (slot m = the state after column 16 m + 15) and every word's
// horizontal output deltas kept as well - two bits per column, 32 columns per u64 {P bits, M bits << 32}, the bit of column 32 m + x at
// position 31 - x (one v_alignbit per plane and step) - so that ANY word can later be recomputed exactly from a checkpoint alone.
// ---- where a checkpoint / a delta word lives.  Slot `slot` (of `slots` per block), word w of block x (work index minus the launch's `lo`):
// the blocks whose word-w lanes store in ONE instruction of the checkpoint pass - the 8 blocks of a list-A wave, the 4 of a list-B wave -
// sit side by side, so that instruction writes one 128- (64-) byte line instead of 16 bytes in 8 (4) lines 4 KB apart: 5 - 7 % of the pass
// (tools/ck_microbench.hip).  The walk reads a block's two words of a slot as two 16-byte pieces either way.
template <int NW> struct RcLay { static constexpr int kGI = NW == 8 ? 8 : NW == 13 ? 4 : 1; };       // blocks per group (the 2048-bp geometries: one block per wave)
template <int NW>
NECAT_D size_t rc_at(u64 x, int slots, size_t slot, size_t w)
{
    constexpr u64 GI = RcLay<NW>::kGI;
    return (size_t)((((x / GI) * (u64)slots + slot) * NW + w) * GI + x % GI);
}
template <int NW> constexpr int kRcStride = NW * RcLay<NW>::kGI;          // elements between two slots' same word

template <int TW, bool CARRY>
NECAT_D u32 fast_shw8_ck(const int b, const u64* __restrict__ tw, const u64 nlo, const u64 nhi, ulonglong2* __restrict__ ck, u64* __restrict__ hc)
{
    constexpr int G = 8, N = kOcaBlockSize, kSteps = N + G - 1;
    const u32 cm = b == G - 1 ? 0x80000000u : 0u;
    const u32 nlo_l = (u32)nlo, nlo_h = (u32)(nlo >> 32), nhi_l = (u32)nhi, nhi_h = (u32)(nhi >> 32);
    const u32 sk = (u32)(32 - b) & 31u;
    const int jck = (b + 31) & 31;               // the step (mod 32) at which this lane's column is 31 mod 32
    const int jck16 = (b + 15) & 15;             // .. (mod 16) at which it is 15 mod 16
    u32 hp = 0, hm = 0;
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
                if (CARRY) {
                    hp = __builtin_amdgcn_alignbit(hp, phh, 31); hm = __builtin_amdgcn_alignbit(hm, mhh, 31);      // (h << 1) | top bit
                    if ((j & 15) == jck16) {
                        ck[(size_t)((s - b) >> 4) * kRcStride<G>] = make_ulonglong2(w.Pv, w.Mv);
                        if (j == jck) hc[(size_t)((s - b) >> 5) * kRcStride<G>] = (u64)hp | ((u64)hm << 32);
                    }
                } else
                if (j == jck) ck[(size_t)((s - b) >> 5) * kRcStride<G>] = make_ulonglong2(w.Pv, w.Mv);      // state after column 32 m + 31 -> slot m
            }
        }
    }
    return key;
}

// fast_shw8_ck<TW, true> with the bottom row's minimum found AFTER the pass.  The DP kernel is bound by VALU issue (88 % of the SIMD cycles at 41
// vector instructions per step, profiles/r04_sq_counters.json), and 8 of those 41 were not the recurrence:
//   * the running score and its minimum (5 per step, on every lane, for a value only word 7 needs): the bottom row's deltas are the very bits
//     word 7 writes to `hc`, so after the pass the 8 lanes of a block read them back - lane x the 64 columns [64 x, 64 x + 64) - start from 512 +
//     the deltas of the columns before (popcounts, a prefix over the 8 lanes) and walk their 64 columns: 5 x 64 instructions instead of 5 x 519;
//   * the per-lane test "is this my checkpoint step" (2): with the 32 steps of a window unrolled the step number is a constant, the lanes
//     that store at it are those of ONE word, and the compare against that word is loop invariant (a scalar mask);
//   * the copy of the new Pv over the old one (1): gone with the unrolling.
// The first and the last window (steps 0 - 31: lanes join one by one; 512 - 518: they leave) keep the rolled, masked loop.
// Returns, on all 8 lanes of the block, what fast_shw8_ck returns on lane 7: (smallest bottom-row value << 10) + the step of word 7 at the
// FIRST column attaining it (= that column + 7).
// fast_advance<false> with the -1 carry published as the NUMBER 0 / 1 (pubM) instead of in bit 31: the word below ORs it into Eq and into
// (Mh << 1) as it comes - one shift per step less; `mw` = 1, or 0 on the lane of word 7, which publishes "no carry" for the top word of
// the next block of its DPP row (v_bfe_u32 with a width of 0 bits)
NECAT_D void fast_advance_m1(FastWord& w, u32 el, u32 eh, u32 cph, u32 cm1, u32 cm, u32 mw, u32& phh_out, u32& mhh_out)
{
    const u32 pl = (u32)w.Pv, ph = (u32)(w.Pv >> 32), ml = (u32)w.Mv, mh = (u32)(w.Mv >> 32);
    const u32 xvl = el | ml, xvh = eh | mh;
    const u32 e2l = el | cm1;
    const u64 sum = (((u64)(eh & ph) << 32) | (e2l & pl)) + w.Pv;
    const u32 sl = (u32)sum, sh = (u32)(sum >> 32);
    const u32 xhl = bop<0xbe>(sl, pl, e2l), xhh = bop<0xbe>(sh, ph, eh);
    const u32 phl = bop<0xf1>(ml, xhl, pl), phh = bop<0xf1>(mh, xhh, ph);
    const u32 mhl = pl & xhl, mhh = ph & xhh;
    phh_out = phh; mhh_out = mhh;
    w.pubP = phh | cm; w.pubM = __builtin_amdgcn_ubfe(mhh, 31u, mw);
    const u32 p2l = __builtin_amdgcn_alignbit(phl, cph, 31), p2h = __builtin_amdgcn_alignbit(phh, phl, 31);
    const u32 m2l = (mhl << 1) | cm1, m2h = __builtin_amdgcn_alignbit(mhh, mhl, 31);
    const u32 ol = bop<0xf1>(m2l, xvl, p2l), oh = bop<0xf1>(m2h, xvh, p2h);
    const u32 nl = p2l & xvl, nh = p2h & xvh;
    w.Pv = ((u64)oh << 32) | ol; w.Mv = ((u64)nh << 32) | nl;
}

template <int TW, int ST = kRcStride<8>>          // ST: distance of two slots' same word in `ck` / `hc` (elements)
NECAT_D u32 fast_shw8_ckp(const int b, const u64* __restrict__ tw, const u64 nlo, const u64 nhi, ulonglong2* __restrict__ ck, u64* __restrict__ hc, const u32 dbg = 0u)
{
    // dbg (tools/ck_microbench.hip only; 0 in the library): bit 0 = no checkpoint / delta stores
    const bool st = !(dbg & 1u);
    constexpr int G = 8, N = kOcaBlockSize, kSteps = N + G - 1;
    static_assert(N == 512 && TW * 32 >= N, "16 windows of 32 columns");
    const u32 cm = b == G - 1 ? 0x80000000u : 0u;
    const u32 nlo_l = (u32)nlo, nlo_h = (u32)(nlo >> 32), nhi_l = (u32)nhi, nhi_h = (u32)(nhi >> 32);
    const u32 sk = (u32)(32 - b) & 31u;
    const int jck = (b + 31) & 31, jck16 = (b + 15) & 15;
    u32 hp = 0, hm = 0;
    u32 tlo = 0, thi = 0, plo = 0, phi = 0;
    FastWord w; w.Pv = ~0ULL; w.Mv = 0ULL; w.pubP = 0x80000000u; w.pubM = 0u;
    const u32 mw = b == G - 1 ? 0u : 1u;
    u32 cph = 0x80000000u, cmh = 0u;           // (cmh: 0 / 1 here, fast_advance_m1)
    auto step = [&](const int j) {          // one column of this lane's word: the target bit of step j of the window, the recurrence, the delta bits
        const u32 ma = (u32)__builtin_amdgcn_sbfe((int)tlo, (u32)j, 1u), mb = (u32)__builtin_amdgcn_sbfe((int)thi, (u32)j, 1u);
        const u32 el = bop<0x60>(nlo_l ^ ma, nhi_l, mb), eh = bop<0x60>(nlo_h ^ ma, nhi_h, mb);
        u32 phh, mhh;
        fast_advance_m1(w, el, eh, cph, cmh, cm, mw, phh, mhh);
        hp = __builtin_amdgcn_alignbit(hp, phh, 31); hm = __builtin_amdgcn_alignbit(hm, mhh, 31);
    };
    // ckr / hcr: the slots of the window's first column (two checkpoints, one delta word per window)
    ulonglong2* ckr = ck; u64* hcr = hc;
    for (int s0 = 0; s0 < kSteps; s0 += 32, ckr += 2 * ST, hcr += ST) {
        {
            const u64 x = (s0 >> 5) < TW ? tw[s0 >> 5] : 0ULL;
            const u32 xl = (u32)x, xh = (u32)(x >> 32);
            tlo = b ? __builtin_amdgcn_alignbit(xl, plo, sk) : xl;
            thi = b ? __builtin_amdgcn_alignbit(xh, phi, sk) : xh;
            plo = xl; phi = xh;
        }
        if (s0 != 0 && s0 + 32 <= N) {
            // steps 32 .. 511: every lane is inside its block; the slots relative to the window's are constants
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                cph = dpp_row_shr1(w.pubP, cph); cmh = dpp_row_shr1(w.pubM, cmh);
                step(j);
                const int K = ((j & 15) + 1) & 15;                 // the word whose column is 15 mod 16 at this step
                if (K < G && b == K && st) {
                    ckr[((j - K) >> 4) * ST] = make_ulonglong2(w.Pv, w.Mv);
                    if (j == ((K + 31) & 31)) hcr[((j - K) >> 5) * ST] = (u64)hp | ((u64)hm << 32);
                }
            }
        } else {
            // (round 6 tried steps 7 .. 31 of the first window unrolled too - 25 of the pass's 39 rolled steps: the kernel grew by a third and its launches by 1 %,
            // tools/r06/run19.sh)
            const int jn = kSteps - s0 < 32 ? kSteps - s0 : 32;
            for (int j = 0; j < jn; ++j) {
                const int s = s0 + j;
                cph = dpp_row_shr1(w.pubP, cph); cmh = dpp_row_shr1(w.pubM, cmh);
                const bool edge = s < G - 1 || s >= N;
                if (!edge || (s >= b && s - b < N)) {
                    step(j);
                    if ((j & 15) == jck16) {
                        ckr[((j - b) >> 4) * ST] = make_ulonglong2(w.Pv, w.Mv);
                        if (j == jck) hcr[((j - b) >> 5) * ST] = (u64)hp | ((u64)hm << 32);
                    }
                }
            }
        }
    }
    // ---- the bottom row: word 7's deltas, 64 columns per lane.  (The stores above and these loads are by lanes of ONE wave: made visible
    // by a workgroup-scope release / acquire - no cache maintenance on gfx950, the CU's vector cache is coherent for its own wavefronts.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const u64* hb = hc + (G - 1 - b) * (ST / G);                      // slot m of word 7: hb[m * ST]
    const u64 g0 = __builtin_nontemporal_load(hb + (size_t)(2 * b) * ST), g1 = __builtin_nontemporal_load(hb + (size_t)(2 * b + 1) * ST);
    const u32 P0 = (u32)g0, M0 = (u32)(g0 >> 32), P1 = (u32)g1, M1 = (u32)(g1 >> 32);
    const int d = (int)__popc(P0) + (int)__popc(P1) - (int)__popc(M0) - (int)__popc(M1);
    int incl = d;
#pragma unroll
    for (int o = 1; o < G; o <<= 1) { const int v = __shfl_up(incl, o); if (b >= o) incl += v; }
    u32 S = (u32)(N + incl - d);                                     // the bottom row's value left of this lane's first column
    u32 key = 0xffffffffu;
    const u32 c0 = (u32)(64 * b + G - 1);
#pragma unroll
    for (int x = 0; x < 32; ++x) {
        S += ((P0 >> (31 - x)) & 1u) - ((M0 >> (31 - x)) & 1u);
        const u32 k2 = (S << 10) + c0 + (u32)x;
        key = k2 < key ? k2 : key;
    }
#pragma unroll
    for (int x = 0; x < 32; ++x) {
        S += ((P1 >> (31 - x)) & 1u) - ((M1 >> (31 - x)) & 1u);
        const u32 k2 = (S << 10) + c0 + 32u + (u32)x;
        key = k2 < key ? k2 : key;
    }
#pragma unroll
    for (int o = 1; o < G; o <<= 1) { const u32 v = (u32)__shfl_xor((int)key, o); key = v < key ? v : key; }
    return key;
}

// fast_shw8_ck<CARRY = true> for a block of ANY size up to 512 x 512 (the ragged blocks of list A: a last block of an extension, qn x tn): the
// same 8-lane wavefront - lane = 64-row word, DPP carries, v_bitop3 logic, one 64-bit add - with what the general pass k_myers_ckg does for
// such a block: only the words the query has (b < nblk), the rows past the query in its last word wildcards (build_peq's pad bits,
// edlib_ex.c:46), only the columns the target has (c < tn), and the distance read where it is - at row qn - 1, a bit inside the last word,
// for every column: the first column with the smallest value, which is what the reference's pad-row shift (:199) and its scan of the last
// W cells (:205-219) together find (a value of a padded column c < W is at least qn, above any cutoff).  Same checkpoints, same deltas,
// the last, partial 32-column group of deltas left-aligned as k_myers_ckg leaves it.  `steps`: the wave's longest tn + nblk - 1.
// (G lanes per block - 8: list A, 16: list B at 13 words; NWS: words per checkpoint / delta slot of the geometry; the key keeps the step in 10 bits)
// FASTW: the unrolled, mask-free form of the windows in which every lane of the wave is inside its block exists in this instance (k_myers_ckf: list B).  NOT in
// k_myers_ck's ragged waves: that kernel is held at 64 registers for its full blocks' 8 waves per SIMD, and with the unrolled windows inside it the full blocks' launches
// were 10 % longer (0.48 -> 0.53 ms on average, spills 76 -> 112 bytes, a third more code) for a ragged path that is 12 % of its blocks (tools/r06/run19.sh).
template <int G, int NWS, int TW, bool FASTW>
NECAT_D u32 fast_shw_ckr(const int b, const int qn, const int tn, const int steps, const u64* __restrict__ tw, const u64 nlo, const u64 nhi,
                         ulonglong2* __restrict__ ck, u64* __restrict__ hc, const bool ckr_fast_windows = true)
{
    static_assert(G == 8 || G == 16, "the carries run on DPP row_shr: a block is half a row or a row of 16 lanes");
    static_assert(TW * 32 + G <= 1024, "step count in 10 bits of the key");
    const int nblk = (qn + 63) >> 6, W = nblk * 64 - qn;
    const bool have = b < nblk, lastw = b == nblk - 1;
    const u64 pad = (lastw && W > 0) ? (~0ULL << ((64 - W) & 63)) : 0ULL;
    const u32 pad_l = (u32)pad, pad_h = (u32)(pad >> 32);
    const int pb = (qn - 1) & 63;
    const bool row_hi = pb >= 32;
    const u32 psh = (u32)pb & 31u;
    const u32 cm = (G == 8 && b == G - 1) ? 0x80000000u : 0u;          // (half a row: the last word publishes the top-row boundary for the block in the other half)
    const u32 nlo_l = (u32)nlo, nlo_h = (u32)(nlo >> 32), nhi_l = (u32)nhi, nhi_h = (u32)(nhi >> 32);
    const u32 sk = (u32)(32 - b) & 31u;
    const int jck = (b + 31) & 31, jck16 = (b + 15) & 15;
    u32 hp = 0, hm = 0;
    u32 tlo = 0, thi = 0, plo = 0, phi = 0;
    FastWord w; w.Pv = ~0ULL; w.Mv = 0ULL; w.pubP = 0x80000000u; w.pubM = 0u;
    u32 S = (u32)qn, key = 0xffffffffu;
    u64 dA, dB;
    u32 cph = 0x80000000u, cmh = 0u;
    constexpr int ST = kRcStride<NWS>;
    for (int s0 = 0; s0 < steps; s0 += 32) {
        {
            const u64 x = (s0 >> 5) < TW ? tw[s0 >> 5] : 0ULL;
            const u32 xl = (u32)x, xh = (u32)(x >> 32);
            tlo = b ? __builtin_amdgcn_alignbit(xl, plo, sk) : xl;
            thi = b ? __builtin_amdgcn_alignbit(xh, phi, sk) : xh;
            plo = xl; phi = xh;
        }
        // Round 6: a window in which EVERY lane of the wave that has a word stays inside its block's columns - and short of its last one - for all 32 steps (the blocks
        // of a wave are of like size: all but the first and the last two or three windows of a list-B wave) runs unrolled, without a per-step lane mask, as the
        // full blocks' pass does (fast_shw8_ckp): the step number is a constant, the lanes that store a checkpoint at it are those of ONE word - a loop-invariant
        // scalar mask -, the running minimum is kept by every lane on a row of its own (only the last word's is read) instead of under an exec mask.  Lanes without a
        // word run along: they store nothing, and nobody reads what they publish (the lane below them has no word either; the last lane of a half row always
        // publishes the boundary).  49 -> 38 vector and 18 -> 2 scalar instructions per step.
        if (FASTW && ckr_fast_windows && __all(!have || (s0 >= b && s0 + 31 - b < tn - 1))) {
            ulonglong2* const ckr = ck + (size_t)(s0 >> 4) * ST;
            u64* const hcr = hc + (size_t)(s0 >> 5) * ST;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                cph = dpp_row_shr1(w.pubP, cph); cmh = dpp_row_shr1(w.pubM, cmh);
                const u32 ma = (u32)__builtin_amdgcn_sbfe((int)tlo, (u32)j, 1u), mb = (u32)__builtin_amdgcn_sbfe((int)thi, (u32)j, 1u);
                const u32 el = bop<0x60>(nlo_l ^ ma, nhi_l, mb) | pad_l, eh = bop<0x60>(nlo_h ^ ma, nhi_h, mb) | pad_h;
                u32 phh, mhh, phl, mhl;
                fast_advance<false>(w, el, eh, cph, cmh, cm, phh, mhh, dA, dB, &phl, &mhl);
                const u32 pw = row_hi ? phh : phl, mw = row_hi ? mhh : mhl;
                S += ((pw >> psh) & 1u) - ((mw >> psh) & 1u);
                const u32 k2 = (S << 10) + (u32)(s0 + j);
                key = k2 < key ? k2 : key;
                asm volatile("" : "+v"(key));                      // (the minimum is taken HERE: left to itself the scheduler sinks the 32 steps' updates below the window and keeps every step's Ph / Mh words alive for them - 188 registers)
                hp = __builtin_amdgcn_alignbit(hp, phh, 31); hm = __builtin_amdgcn_alignbit(hm, mhh, 31);
                const int K = ((j & 15) + 1) & 15;                 // the word whose column is 15 mod 16 at this step
                if (K < G && b == K && have) {
                    ckr[((j - K) >> 4) * ST] = make_ulonglong2(w.Pv, w.Mv);
                    if (j == ((K + 31) & 31)) hcr[((j - K) >> 5) * ST] = (u64)hp | ((u64)hm << 32);
                }
            }
            continue;
        }
        const int jn = steps - s0 < 32 ? steps - s0 : 32;
        for (int j = 0; j < jn; ++j) {
            const int s = s0 + j, c = s - b;
            cph = dpp_row_shr1(w.pubP, cph); cmh = dpp_row_shr1(w.pubM, cmh);
            if (have && (u32)c < (u32)tn) {
                const u32 ma = (u32)__builtin_amdgcn_sbfe((int)tlo, (u32)j, 1u), mb = (u32)__builtin_amdgcn_sbfe((int)thi, (u32)j, 1u);
                const u32 el = bop<0x60>(nlo_l ^ ma, nhi_l, mb) | pad_l, eh = bop<0x60>(nlo_h ^ ma, nhi_h, mb) | pad_h;
                u32 phh, mhh, phl, mhl;
                fast_advance<false>(w, el, eh, cph, cmh, cm, phh, mhh, dA, dB, &phl, &mhl);
                if (lastw) {
                    const u32 pw = row_hi ? phh : phl, mw = row_hi ? mhh : mhl;
                    S += ((pw >> psh) & 1u) - ((mw >> psh) & 1u);
                    const u32 k2 = (S << 10) + (u32)s;
                    key = k2 < key ? k2 : key;
                }
                hp = __builtin_amdgcn_alignbit(hp, phh, 31); hm = __builtin_amdgcn_alignbit(hm, mhh, 31);
                if ((j & 15) == jck16) {
                    ck[(size_t)(c >> 4) * kRcStride<NWS>] = make_ulonglong2(w.Pv, w.Mv);
                    if (j == jck) hc[(size_t)(c >> 5) * kRcStride<NWS>] = (u64)hp | ((u64)hm << 32);
                }
                if (c == tn - 1 && (c & 31) != 31) { const int sh = 31 - (c & 31); hc[(size_t)(c >> 5) * kRcStride<NWS>] = (u64)(hp << sh) | ((u64)(hm << sh) << 32); }
            }
        }
    }
    return key;
}

// the front part of list A (work indices [0, nf): full blocks; [nf, nf16): holes), 8 items per wave; flags bit 27 (CARRY only): the whole
// list - the ragged blocks at its back too, their waves (and the one wave that may hold both kinds) through fast_shw_ckr
#ifndef NECAT_CK_WAVES
#define NECAT_CK_WAVES 8          // (tools/ck_microbench.hip builds it with 7 and 6 as well: more registers, fewer waves)
#endif
template <int NW, int TW, bool CARRY>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NECAT_CK_WAVES, NECAT_CK_WAVES)))
k_myers_ck(const BlockItem* __restrict__ items, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, ulonglong2* __restrict__ ckpt,
           u64* __restrict__ hcar, double error, BlockResult* __restrict__ results, unsigned long long* __restrict__ stats, int max_dist, u32 lo, u32 hi, u32 flags,
           const u64* __restrict__ q_bases = nullptr, const u64* __restrict__ t_bases = nullptr)
{
    // flags bit 22 (round 6, NECAT_FRAG_FUSE; needs bit 27): the wave cuts its 8 blocks' fragments out of the 2-bit volumes itself (q_bases / t_bases) - what k_ext_frag
    // did in a launch of its own before every pass - and leaves them in `frag` for the walk and the finishing kernel: one kernel (30 - 54 us + its ramp and drain + a
    // launch gap) less in every round's chain.  Same words, same places, same conditions as k_ext_frag.
    // work items [lo, hi) of the list (multiples of 64: a big list goes through a bounded checkpoint buffer in several launches);
    // checkpoint / delta slots are indexed by item - lo
    constexpr int FW = 2 * NW + TW, G = 8, N = kOcaBlockSize;
    if ((flags >> 23) & 1u) __builtin_amdgcn_s_setprio(3);          // (NECAT_RC_PRIO bit 1)
    __shared__ u64 t_lds[8][TW];
    const ListView lv = list_view(0u, n_dev, capA);
    const bool all = CARRY && ((flags >> 27) & 1u) != 0;
    const u64 first = (u64)lo + (u64)blockIdx.x * 8, lim = all ? lv.n : lv.nf, end = lim < hi ? lim : hi;
    if (first >= end) return;
    const int lane = (int)threadIdx.x, sub = lane >> 3, b = lane & 7;
    const u64 item = first + (u64)sub;
    bool valid = item < end;
    int qn = N, tn = N;
    FragGeom geo; geo.q_base = geo.t_base = 0; geo.q_dir = geo.t_dir = 1; geo.q_comp = geo.t_comp = 0;
    if (all && valid) {
        BlockItem it0;
        valid = list_item(lv, items, item, it0);
        if (valid) { qn = it0.qn; tn = it0.tn; geo = it0.g; }
    }
    const bool ragged = all && __any(valid && (qn != N || tn != N));
    const u64 grp = item >> 6;
    const int il = (int)(item & 63);
    const u64* fr = frag + grp * FW * 64 + il;
    const int nblk = (qn + 63) >> 6;
    u64 nlo = 0, nhi = 0;
    if (CARRY && ((flags >> 22) & 1u)) {
        u64* const fw = const_cast<u64*>(fr);
        if (valid && b < nblk) {
            u64 plo, phi;
            load64_planes(q_bases, geo.q_base, geo.q_dir, geo.q_comp, b * 64, &plo, &phi);
            nlo = ~plo; nhi = ~phi;
            fw[(u64)b * 64] = nlo; fw[(u64)(NW + b) * 64] = nhi;
        }
        for (int w = b; w < TW; w += G) {
            u64 x = 0ULL;
            if (valid && w * 32 < tn) { x = load32_dir(t_bases, geo.t_base + (i64)geo.t_dir * (w * 32), geo.t_dir, geo.t_comp); fw[(u64)(2 * NW + w) * 64] = x; }
            t_lds[sub][w] = even_bits(x) | (even_bits(x >> 1) << 32);
        }
    } else {
    if (valid && b < nblk) { nlo = fr[(u64)b * 64]; nhi = fr[(u64)(NW + b) * 64]; }
    for (int w = b; w < TW; w += G) {
        const u64 x = (valid && w * 32 < tn) ? fr[(u64)(2 * NW + w) * 64] : 0ULL;
        t_lds[sub][w] = even_bits(x) | (even_bits(x >> 1) << 32);
    }
    }
    __syncthreads();
    ulonglong2* const ckp = ckpt + rc_at<G>(item - lo, CARRY ? kRcCk16 : kRcCk, 0, (size_t)b);
    u64* const hcp = hcar + rc_at<G>(item - lo, kRcCk, 0, (size_t)b);
    u32 key;
    if (ragged) {
        int steps = valid ? tn + nblk - 1 : 0;
        for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(steps, o); steps = x > steps ? x : steps; }
        key = fast_shw_ckr<8, 8, TW, false>(b, valid ? qn : 0, valid ? tn : 0, steps, t_lds[sub], nlo, nhi, ckp, hcp);
#ifdef NECAT_CK_MICRO
    } else if (CARRY && !((flags >> 24) & 1u)) key = fast_shw8_ckp<TW>(b, t_lds[sub], nlo, nhi, ckp, hcp, (flags >> 20) & 1u);
#else
    } else if (CARRY && !((flags >> 24) & 1u)) key = fast_shw8_ckp<TW>(b, t_lds[sub], nlo, nhi, ckp, hcp);        // (bit 24, NECAT_CK_POST=0: the minimum tracked inside the pass, as before)
#endif
    else key = fast_shw8_ck<TW, CARRY>(b, t_lds[sub], nlo, nhi, ckp, hcp);
    const int bl = ragged ? nblk - 1 : G - 1;                           // the word the distance was read in
    const u32 bkey = (u32)__shfl((int)key, (lane & ~(G - 1)) | bl);
    int best = (int)(bkey >> 10);
    const int end0 = (int)(bkey & 1023u) - bl;
    const int k0 = (int)((double)(qn < tn ? qn : tn) * error * 1.1);     // edlib_ex.c:751
    if (best > k0) best = -1;
    int err = 0;
    if (best >= 0) { int ad = end0 + 1 - qn; if (ad < 0) ad = -ad; if (best < ad) err = 1; }
    const bool full = qn == N && tn == N;
    if (b == G - 1 && valid) {
        BlockResult br; br.dist = err ? -1 : best; br.endc = (full || best >= 0) ? end0 : -1; br.err = err;      // (a ragged block without an end column: -1, as k_myers_ckg leaves it)
        br.words = (u32)(nblk * tn) | ((full && best > max_dist && !err) ? kWideFlag : 0u);      // max_dist <= kRcMaxDist (smaller in tests: more blocks take the old path)
        results[item] = br;
    }
    // the work counters once per wave (its 8 blocks together): they are three words of ONE cache line, and 3 atomics per block were 580 k
    // same-line atomics per launch of a big round
    const bool owner = b == G - 1 && valid;
    const u64 m_all = __ballot(owner), m_walk = __ballot(owner && best >= 0 && !err && !(full && best > max_dist));
    unsigned long long wsum = owner ? (unsigned long long)(nblk * tn) : 0ULL, bsum = owner ? (unsigned long long)(qn + tn) : 0ULL;
    if (ragged) for (int o = 32; o > 0; o >>= 1) { wsum += __shfl_xor(wsum, o); bsum += __shfl_xor(bsum, o); }
    else { wsum = (unsigned long long)popc64(m_all) * (unsigned long long)(NW * N); bsum = (unsigned long long)popc64(m_all) * (unsigned long long)(2 * N); }
    if (lane == 0 && m_all) {
        stat_add(stats, 0, wsum); stat_add(stats, 1, bsum);
        if (m_walk) stat_add(stats, 3, (unsigned long long)popc64(m_walk));           // blocks the recomputing walk will take
    }
}

NECAT_D u32 dpp_quad_shr1(u32 v, u32 keep)      // lane i of every quad receives v of lane i - 1; lane 0 of a quad keeps `keep`
{
    const u32 x = (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x90 /* quad_perm:[0,0,1,2] */, 0xf, 0xf, false);
    return (threadIdx.x & 3u) ? x : keep;
}

// 16 full blocks per wave, 4 lanes per block (see the head of the file).  ops: the walk's ops, op i of work item x at
// ops_pool[(x / 64) * MAXOPS * 64 + i * 64 + (x % 64)] (k_traceback's layout), written while `store` (the task has not yet seen its
// first run of 8 matches, or the caller keeps the columns).
template <int NW, int TW, int MAXOPS>
__global__ void __launch_bounds__(64)
k_rcwalk4(const BlockItem* __restrict__ items, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, const ulonglong2* __restrict__ ckpt,
          const BlockResult* __restrict__ results, const ExtTask* __restrict__ tasks, int keep_cols, int tail_match_len, u8* __restrict__ ops_pool, WalkOut* __restrict__ wout,
          unsigned long long* __restrict__ stats, int* __restrict__ err_flag, u32 lo, u32 hi)
{
    constexpr int FW = 2 * NW + TW, N = kOcaBlockSize, SEG = kRcSeg;
    __shared__ ulonglong2 slices[16][SEG];
    const ListView lv = list_view(0u, n_dev, capA);
    const u64 first = (u64)lo + (u64)blockIdx.x * 16, end = lv.nf < hi ? lv.nf : hi;
    if (first >= end) return;
    const int lane = (int)threadIdx.x, q = lane >> 2, j = lane & 3;
    const u64 item = first + (u64)q;
    const bool valid = item < end;
    const u64 grp = first >> 6;                                      // 16 consecutive work indices share a 64-item group
    const int il = (int)(item & 63);
    const u64* fr = frag + grp * FW * 64 + il;
    int best = -1, endc = -1;
    bool store = false;
    int mlen = kOcaMatCnt;                                           // the run of matches the tail scan looks for (TailScan::M)
    if (valid) {
        const BlockResult br = results[item];
        if (!(br.words & kWideFlag)) { best = br.dist; endc = br.endc; }     // wide blocks: the old path walks them
        if (best >= 0) { const ExtTask& t = tasks[items[item].task]; store = keep_cols || !t.found; if (t.last) mlen = tail_match_len; }      // (ext_block_done: an aligned block ends its extension iff it is the last one)
    }
    // the band of r - c (ext_fast16.h): lo_x <= r - c <= hi_x
    const int tn2 = endc + 1, d = N - tn2, ad = d < 0 ? -d : d, slack = best >= 0 ? (best - ad) >> 1 : 0;
    const int lo_x = (d < 0 ? d : 0) - slack, hi_x = (d > 0 ? d : 0) + slack;
    int r = N - 1, c = endc;
    int n = 0, nmat = 0, m = 0, hit = 0, nq = 0, nt = 0, acnt = 0, qcnt = 0, tcnt = 0, mcnt = 0;
    bool fin = best < 0;
    u8* const ops = ops_pool + (size_t)(item >> 6) * MAXOPS * 64 + il;
    int wcur = -1; u32 nlo_l = 0, nlo_h = 0, nhi_l = 0, nhi_h = 0;      // the query planes of the lane's current word
    int segcur = -1; u32 tlo = 0, thi = 0;                              // the target bit-planes of the current segment's 32 columns
    u32 words_done = 0;
    while (!__all(fin)) {
        const int seg = c >> 5, c0 = seg * SEG;
        const int rb = r - 63;                                        // the walk may use rows [rb, r] of this segment's columns
        int wtop = (c0 + lo_x) >> 6;                                  // floor: the word of the band's top row at the segment's first column
        wtop = wtop < 0 ? 0 : (wtop > NW - 4 ? NW - 4 : wtop);
        const int w = wtop + j;
        if (!fin && w != wcur) {
            const u64 a = fr[(u64)w * 64], bq = fr[(u64)(NW + w) * 64];
            nlo_l = (u32)a; nlo_h = (u32)(a >> 32); nhi_l = (u32)bq; nhi_h = (u32)(bq >> 32); wcur = w;
        }
        FastWord wd; wd.Pv = ~0ULL; wd.Mv = 0ULL; wd.pubP = 0x80000000u; wd.pubM = 0u;
        if (!fin && seg > 0) { const ulonglong2 v = ckpt[rc_at<8>(item - lo, kRcCk, (size_t)(seg - 1), (size_t)w)]; wd.Pv = v.x; wd.Mv = v.y; }
        if (!fin && seg != segcur) {
            const u64 x = fr[(u64)(2 * NW + seg) * 64];
            tlo = (u32)even_bits(x); thi = (u32)even_bits(x >> 1); segcur = seg;
        }
        const int ncol = fin ? 0 : (c - c0 + 1);                      // columns c0 .. c of the segment are needed
        const int wa = rb >> 6, sh = rb & 63, w1 = r >> 6;            // slice rows [rb, r]: word wa from bit sh on, then word w1 = wa + 1 (sh != 0)
        for (int s = 0; s < SEG + 3; ++s) {
            const u32 cph = dpp_quad_shr1(wd.pubP, 0x80000000u), cmh = dpp_quad_shr1(wd.pubM, 0u);      // the window's top word: boundary carry
            const int ci = s - j;
            if ((u32)ci < (u32)ncol) {
                const u32 ma = (u32)__builtin_amdgcn_sbfe((int)tlo, (u32)ci, 1u), mb = (u32)__builtin_amdgcn_sbfe((int)thi, (u32)ci, 1u);
                const u32 el = bop<0x60>(nlo_l ^ ma, nhi_l, mb), eh = bop<0x60>(nlo_h ^ ma, nhi_h, mb);
                u32 phh, mhh; u64 rA, rB;
                fast_advance<true>(wd, el, eh, cph, cmh, 0u, phh, mhh, rA, rB);
                ++words_done;
                // the walk's 64 rows of this column: low part from word wa (written first: its lane is one step ahead), high part
                // from word w1
                if (w == wa) slices[q][ci] = sh ? make_ulonglong2(rA >> sh, rB >> sh) : make_ulonglong2(rA, rB);
                else if (w == w1 && sh) {
                    const u64 pa = rA << (64 - sh), pb = rB << (64 - sh);
                    if (wa < wtop) slices[q][ci] = make_ulonglong2(pa, pb);       // the rows above the window are never looked at
                    else {      // (LDS atomics without a return value: nothing to wait for)
                        unsigned long long* dst = reinterpret_cast<unsigned long long*>(&slices[q][ci]);
                        __hip_atomic_fetch_or(dst, pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_or(dst + 1, pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        __syncthreads();
        // ---- the walk (walk_block of dp_core.h on the slices), every lane of the quad the same.  Two forms of the loop: once every
        // block of the wave has seen its run of 8 matches and no block keeps its ops (every block but the first one or two of an
        // extension, when only records are wanted) a step only counts columns and matches
        const bool lean = !__any(!fin && (!hit || store));
        bool out = false;                                             // walked out of the matrix: the block is done
        if (!fin) {
            if (lean) {
                for (;;) {
                    if (c < c0 || r < rb) break;                      // out of the segment / of the rows kept: the next segment (or this one again)
                    const ulonglong2 v = slices[q][c - c0];
                    const int bit = r - rb;
                    const u32 a = (u32)(v.x >> bit) & 1u, b = (u32)(v.y >> bit) & 1u;
                    const int drow = 1 - (int)(b & (a ^ 1u)), dcol = 1 - (int)(a & (b ^ 1u));
                    ++n; nmat += (int)((a | b) ^ 1u);
                    r -= drow; c -= dcol;
                    if ((r | c) < 0) { out = true; break; }
                }
            } else {
                for (;;) {
                    if (c < c0 || r < rb) break;
                    const ulonglong2 v = slices[q][c - c0];
                    const int bit = r - rb;
                    const u32 a = (u32)(v.x >> bit) & 1u, b = (u32)(v.y >> bit) & 1u;
                    const int op = (int)(a | (b << 1));
                    const int drow = 1 - (int)(b & (a ^ 1u)), dcol = 1 - (int)(a & (b ^ 1u));
                    const int mt = (int)((a | b) ^ 1u);
                    if (store && j == 0) { if (n < MAXOPS) ops[(size_t)n * 64] = (u8)op; else atomicExch(err_flag, 20); }
                    ++n; nmat += mt;
                    if (!hit) {
                        nq += drow; nt += dcol;
                        m = mt ? m + 1 : 0;
                        if (m == mlen) { hit = 1; acnt = n; qcnt = nq; tcnt = nt; mcnt = nmat; }
                    }
                    r -= drow; c -= dcol;
                    if ((r | c) < 0) { out = true; break; }
                }
            }
        }
        if (out) {
            // out of the first column: the rows left are inserts; out of the first row: the columns left are deletes
            const int kop = c < 0 ? 1 : 2, k = c < 0 ? r + 1 : c + 1;
            if (store && j == 0) for (int i = 0; i < k; ++i) { if (n + i < MAXOPS) ops[(size_t)(n + i) * 64] = (u8)kop; else atomicExch(err_flag, 20); }
            n += k;
            if (!hit && k > 0) m = 0;
            fin = true;
            if (j == 0) { WalkOut o; o.n = n; o.nmat = nmat; o.m = m; o.hit = hit; o.acnt = acnt; o.qcnt = qcnt; o.tcnt = tcnt; o.mcnt = mcnt; wout[item] = o; }
        }
        __syncthreads();             // the slices are read before the next segment overwrites them
    }
    for (int o = 32; o > 0; o >>= 1) words_done += (u32)__shfl_xor((int)words_done, o);
    if (lane == 0 && words_done) { stat_add(stats, 0, (unsigned long long)words_done); stat_add(stats, 4, (unsigned long long)words_done); }
}


NECAT_D u32 dpp_quad_from_below(u32 v)          // lane i of every quad receives v of lane i - 1 (lane 0 of a quad: its own)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x90 /* quad_perm:[0,0,1,2] */, 0xf, 0xf, false);
}

// checkpoint / delta slots of a block of up to COLS columns
template <int COLS> struct RcGeom { static constexpr int kCk = (COLS + 15) / 16, kSeg = (COLS + 31) / 32; };

// ---- SHW pass of ANY block (ragged list-A blocks, list B, the 2048-bp geometry) with checkpoints and deltas: the SHW part of
// myers_coop_wave (ext_kernels.h; edlib_ex.c:108-223) - G lanes per block, lane b = 64-row word b - that keeps, as fast_shw8_ck<CARRY> does,
// every word's (Pv, Mv) after columns 15, 31, .. (slot c / 16) and its horizontal output deltas (32 columns per u64, column 32 m + x at
// bit 31 - x of each half).  Work items [lo, hi) of the list; epoch bit 26: only the ragged part [nf16, n) of a two-ended list A.
template <int NW, int TW, int COLS, int G>
__global__ void __launch_bounds__(64)
k_myers_ckg(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, ulonglong2* __restrict__ ckpt,
            u64* __restrict__ hcar, double error, BlockResult* __restrict__ results, unsigned long long* __restrict__ stats, u32 epoch, u32 lo, u32 hi)
{
    constexpr int FW = 2 * NW + TW, BPW = 64 / G, CK = RcGeom<COLS>::kCk, SEGS = RcGeom<COLS>::kSeg;
    __shared__ u64 t_lds[BPW][TW];
    const ListView lv = list_view(n_host, n_dev, capA);
    const u64 wave_first = (u64)lo + (u64)blockIdx.x * BPW, end = lv.n < hi ? lv.n : hi;
    if (wave_first >= end) return;
    if (((epoch >> 26) & 1u) && wave_first + BPW <= (u64)lv.nf16) return;
    const int lane = (int)threadIdx.x, sub = lane / G, b = lane % G;
    const u64 item = wave_first + (u64)sub;
    BlockItem it0;
    bool valid = item < end && list_item(lv, items, item, it0);
    if (((epoch >> 26) & 1u) && item < (u64)lv.nf16) valid = false;
    const u64 grp = item >> 6;
    const int il = (int)(item & 63);
    int qn = 0, tn = 0;
    if (valid) { qn = it0.qn; tn = it0.tn; }
    const int nblk = (qn + 63) >> 6, W = nblk * 64 - qn;
    const bool have = valid && b < nblk;
    const bool is_last = have && b == nblk - 1;
    const u64* fr = frag + grp * FW * 64 + il;
    u64 nlo = 0, nhi = 0;
    if (have) { nlo = fr[(u64)b * 64]; nhi = fr[(u64)(NW + b) * 64]; }
    const u64 pad = (is_last && W > 0) ? (~0ULL << ((64 - W) & 63)) : 0ULL;
    if (valid) for (int w = b; w < TW; w += G) {
        const u64 x = (w * 32 < tn) ? fr[(u64)(2 * NW + w) * 64] : 0ULL;
        t_lds[sub][w] = even_bits(x) | (even_bits(x >> 1) << 32);
    }
    __syncthreads();
    const u64* tw = t_lds[sub];
    const u32 nlo_l = (u32)nlo, nlo_h = (u32)(nlo >> 32), nhi_l = (u32)nhi, nhi_h = (u32)(nhi >> 32);
    const u32 pad_l = (u32)pad, pad_h = (u32)(pad >> 32);
    u64 tcur = 0;
    auto eq_of = [&](int c) -> u64 {
        const u32 ma = (u32)__builtin_amdgcn_sbfe((int)(u32)tcur, (u32)c & 31u, 1u);
        const u32 mb = (u32)__builtin_amdgcn_sbfe((int)(u32)(tcur >> 32), (u32)c & 31u, 1u);
        const u32 el = ((nlo_l ^ ma) & (nhi_l ^ mb)) | pad_l, eh = ((nlo_h ^ ma) & (nhi_h ^ mb)) | pad_h;
        return ((u64)eh << 32) | el;
    };
    int steps = valid ? tn + nblk - 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(steps, o); steps = x > steps ? x : steps; }
    ulonglong2* const ck = ckpt + rc_at<NW>(item - lo, CK, 0, (size_t)b);
    u64* const hc = hcar + rc_at<NW>(item - lo, SEGS, 0, (size_t)b);
    int k = (int)((double)(qn < tn ? qn : tn) * error * 1.1);
    u64 P = ~0ULL, M = 0ULL;
    int S = (b + 1) * 64, best = -1, end0 = -1, hout = 1;
    u32 hp = 0, hm = 0;
    for (int s = 0; s < steps; ++s) {
        const int c = s - b;
        int hin = lane_below<G>(hout);
        if (b == 0) hin = 1;
        if (have && (u32)c < (u32)tn) {
            if ((c & 31) == 0) tcur = tw[c >> 5];
            const u64 eq = eq_of(c);
            u64 rA, rB;
            hout = advance_dev<false>(P, M, eq, hin, rA, rB);
            S += hout;
            hp = (hp << 1) | ((u32)(hout + 1) >> 1); hm = (hm << 1) | ((u32)hout >> 31);
            if ((c & 15) == 15) {
                ck[(size_t)(c >> 4) * kRcStride<NW>] = make_ulonglong2(P, M);
                if ((c & 31) == 31) hc[(size_t)(c >> 5) * kRcStride<NW>] = (u64)hp | ((u64)hm << 32);
            }
            if (c == tn - 1 && (c & 31) != 31) { const int sh = 31 - (c & 31); hc[(size_t)(c >> 5) * kRcStride<NW>] = (u64)(hp << sh) | ((u64)(hm << sh) << 32); }
            if (is_last && S <= k && (best == -1 || S <= best)) {
                if (S != best) { best = S; k = best; end0 = c - W; }
            }
        }
    }
    if (is_last && W > 0) {          // edlib_ex.c:205-219
        int score = S;
        for (int i = 0; i < W; ++i) {
            if (P & (kHighBit >> i)) --score;
            if (M & (kHighBit >> i)) ++score;
            if (score <= k && (best == -1 || score <= best)) {
                if (score != best) { k = best = score; end0 = tn - W + i; }
            }
        }
    }
    if (is_last) {
        const int tn2 = end0 + 1;
        int err = 0;
        if (best >= 0) { int ad = tn2 - qn; if (ad < 0) ad = -ad; if (best < ad) err = 1; }
        BlockResult br; br.dist = err ? -1 : best; br.endc = end0; br.err = err;
        br.words = (u32)(nblk * tn);
        results[item] = br;
    }
    {   // the work counters once per wave
        unsigned long long w = is_last ? (unsigned long long)(nblk * tn) : 0ULL, bs = is_last ? (unsigned long long)(qn + tn) : 0ULL;
        for (int o = 32; o > 0; o >>= 1) { w += __shfl_xor(w, o); bs += __shfl_xor(bs, o); }
        if (lane == 0 && w) { stat_add(stats, 0, w); stat_add(stats, 1, bs); }
    }
}

// k_myers_ckg's drop-in for blocks of at most 16 words (list B: 13 words, 16 lanes per block, four blocks per wave): the same results, checkpoints
// and deltas through fast_shw_ckr - 32-bit halves, v_bitop3 logic, DPP carries - instead of the general pass's 64-bit steps.
template <int NW, int TW, int COLS, int G>
__global__ void __launch_bounds__(64)
k_myers_ckf(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, ulonglong2* __restrict__ ckpt,
            u64* __restrict__ hcar, double error, BlockResult* __restrict__ results, unsigned long long* __restrict__ stats, u32 epoch, u32 lo, u32 hi)
{
    constexpr int FW = 2 * NW + TW, BPW = 64 / G, CK = RcGeom<COLS>::kCk, SEGS = RcGeom<COLS>::kSeg;
    static_assert(NW <= G, "one lane per word");
    __shared__ u64 t_lds[BPW][TW];
    const ListView lv = list_view(n_host, n_dev, capA);
    const u64 wave_first = (u64)lo + (u64)blockIdx.x * BPW, end = lv.n < hi ? lv.n : hi;
    if (wave_first >= end) return;
    if (((epoch >> 26) & 1u) && wave_first + BPW <= (u64)lv.nf16) return;
    const int lane = (int)threadIdx.x, sub = lane / G, b = lane % G;
    const u64 item = wave_first + (u64)sub;
    BlockItem it0;
    bool valid = item < end && list_item(lv, items, item, it0);
    if (((epoch >> 26) & 1u) && item < (u64)lv.nf16) valid = false;
    const u64 grp = item >> 6;
    const int il = (int)(item & 63);
    const int qn = valid ? it0.qn : 0, tn = valid ? it0.tn : 0;
    const int nblk = (qn + 63) >> 6;
    const u64* fr = frag + grp * FW * 64 + il;
    u64 nlo = 0, nhi = 0;
    if (valid && b < nblk) { nlo = fr[(u64)b * 64]; nhi = fr[(u64)(NW + b) * 64]; }
    if (valid) for (int w = b; w < TW; w += G) {
        const u64 x = (w * 32 < tn) ? fr[(u64)(2 * NW + w) * 64] : 0ULL;
        t_lds[sub][w] = even_bits(x) | (even_bits(x >> 1) << 32);
    }
    __syncthreads();
    int steps = valid ? tn + nblk - 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(steps, o); steps = x > steps ? x : steps; }
    const u32 key = fast_shw_ckr<G, NW, TW, true>(b, qn, tn, steps, t_lds[sub], nlo, nhi, ckpt + rc_at<NW>(item - lo, CK, 0, (size_t)b), hcar + rc_at<NW>(item - lo, SEGS, 0, (size_t)b),
                                            !((epoch >> 28) & 1u));      // (bit 28, NECAT_CKR_FAST=0: every window rolled, as until round 5)
    const int bl = nblk > 0 ? nblk - 1 : 0;
    const u32 bkey = (u32)__shfl((int)key, (lane / G) * G + bl);
    int best = (int)(bkey >> 10);
    const int end0 = (int)(bkey & 1023u) - bl;
    const int k0 = (int)((double)(qn < tn ? qn : tn) * error * 1.1);     // edlib_ex.c:751
    if (bkey == 0xffffffffu || best > k0) best = -1;
    int err = 0;
    if (best >= 0) { int ad = end0 + 1 - qn; if (ad < 0) ad = -ad; if (best < ad) err = 1; }
    const bool owner = valid && b == 0;
    if (owner) {
        BlockResult br; br.dist = err ? -1 : best; br.endc = best >= 0 ? end0 : -1; br.err = err;
        br.words = (u32)(nblk * tn);
        results[item] = br;
    }
    {   // the work counters once per wave
        unsigned long long w = owner ? (unsigned long long)(nblk * tn) : 0ULL, bs = owner ? (unsigned long long)(qn + tn) : 0ULL;
        for (int o = 32; o > 0; o >>= 1) { w += __shfl_xor(w, o); bs += __shfl_xor(bs, o); }
        if (lane == 0 && w) { stat_add(stats, 0, w); stat_add(stats, 1, bs); }
    }
}

// k_rcwalk4 with an EXACT window of two words (needs the deltas: k_myers_ck<.., CARRY = true> / k_myers_ckg): the walk only looks at rows
// [r - 63, r] (r = its row when the segment is entered) - the words w1 = r / 64 and w1 - 1 - and a word's column is a function of its own
// previous column, the word above's horizontal deltas and the sequences; with those deltas on record (hcar) two words are all there is to
// compute, for any block size and any distance: no band argument, no `wide` blocks, no pad rows (a row never depends on a row below it).
// 4 lanes per block = 2 words x 2 halves of the 32-column segment (checkpoints every 16 columns): 17 steps per segment instead of 35.
// Work items [lo, hi) of any list (ListView); epoch bit 27: the whole list (otherwise only the full blocks [0, nf) of a two-ended list A).
template <int NW, int TW, int COLS, int MAXOPS>
__global__ void __launch_bounds__(64)
k_rcwalk2(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, const ulonglong2* __restrict__ ckpt,
          const u64* __restrict__ hcar, const BlockResult* __restrict__ results, const ExtTask* __restrict__ tasks, int keep_cols, int tail_match_len, u8* __restrict__ ops_pool,
          WalkOut* __restrict__ wout, unsigned long long* __restrict__ stats, int* __restrict__ err_flag, u32 epoch, u32 lo, u32 hi)
{
    constexpr int FW = 2 * NW + TW, SEG = kRcSeg, HALF = SEG / 2, CK = RcGeom<COLS>::kCk, SEGS = RcGeom<COLS>::kSeg;
    __shared__ ulonglong2 slices[SEG][16];      // [column][block]: a block's 16 bytes of every column sit in its own 4 LDS banks - blocks never conflict, whatever column each is at
    const ListView lv = list_view(n_host, n_dev, capA);
    // epoch bit 27: the whole list; bit 26: only the ragged part [nf16, n) of a two-ended list A; neither: only its full blocks [0, nf)
    const bool all = ((epoch >> 27) & 1u) != 0, ragged = ((epoch >> 26) & 1u) != 0;
    const u64 first = (u64)lo + (u64)blockIdx.x * 16, lim = (all || ragged) ? lv.n : lv.nf, end = lim < hi ? lim : hi;
    if (first >= end || (ragged && first + 16 <= (u64)lv.nf16)) return;      // (nf16 is a multiple of 16: a wave is all full blocks / holes or all ragged ones)
    const int lane = (int)threadIdx.x, q = lane >> 2, j = lane & 3, k = j & 1, h = j >> 1;
    const u64 item = first + (u64)q;
    BlockItem it0;
    const bool valid = item < end && list_item(lv, items, item, it0);
    const u64 grp = first >> 6;                                      // 16 consecutive work indices share a 64-item group
    const int il = (int)(item & 63);
    const u64* fr = frag + grp * FW * 64 + il;
    int best = -1, endc = -1;
    bool store = false;
    int mlen = kOcaMatCnt;                                           // the run of matches the tail scan looks for (TailScan::M)
    if (valid) {
        const BlockResult br = results[item];
        if (!(br.words & kWideFlag)) { best = br.dist; endc = br.endc; }     // (flagged only when a test lowers NECAT_RC_MAXDIST: the old path walks them)
        // (ext_block_done: an aligned block ends its extension iff it is the last one; no tasks: the batch hook, which keeps every op)
        if (best >= 0 && tasks) { const ExtTask& t = tasks[it0.task]; store = keep_cols || !t.found; if (t.last) mlen = tail_match_len; }
        else if (best >= 0) store = true;
    }
    int r = valid ? it0.qn - 1 : 0, c = endc;
    int n = 0, nmat = 0, m = 0, hit = 0, nq = 0, nt = 0, acnt = 0, qcnt = 0, tcnt = 0, mcnt = 0;
    bool fin = best < 0;
    u8* const ops = ops_pool + (size_t)(item >> 6) * MAXOPS * 64 + il;
    int wcur = -1; u32 nlo_l = 0, nlo_h = 0, nhi_l = 0, nhi_h = 0;      // the query planes of the lane's current word
    int segcur = -1; u32 tlo = 0, thi = 0;                              // the target bit-planes of the current segment's 32 columns
    u32 words_done = 0;
    while (!__all(fin)) {
        const int seg = c >> 5, c0 = seg * SEG;
        const int rb = r - 63, sh = rb & 63, w1 = r >> 6;            // the walk may use rows [rb, r] of this segment's columns: word w1 - 1 from bit sh on, word w1
        const int w = w1 - 1 + k;
        const int nc0 = c - c0 - HALF * h + 1;                        // columns of this lane's half that are needed
        const int nc = (fin || w < 0 || nc0 < 0) ? 0 : (nc0 > HALF ? HALF : nc0);
        const bool live = nc > 0;
        if (live && w != wcur) {
            const u64 a = fr[(u64)w * 64], bq = fr[(u64)(NW + w) * 64];
            nlo_l = (u32)a; nlo_h = (u32)(a >> 32); nhi_l = (u32)bq; nhi_h = (u32)(bq >> 32); wcur = w;
        }
        FastWord wd; wd.Pv = ~0ULL; wd.Mv = 0ULL; wd.pubP = 0x80000000u; wd.pubM = 0u;
        const int slot = 2 * seg + h - 1;                             // the state before column c0 + 16 h
        if (live && slot >= 0) { const ulonglong2 v = ckpt[rc_at<NW>(item - lo, CK, (size_t)slot, (size_t)w)]; wd.Pv = v.x; wd.Mv = v.y; }
        u32 hp = 0xffffffffu, hm = 0u;                                // word 0: the top row's boundary (+1 per column)
        if (live && k == 0 && w > 0) { const u64 v = hcar[rc_at<NW>(item - lo, SEGS, (size_t)seg, (size_t)(w - 1))]; hp = (u32)v; hm = (u32)(v >> 32); }
        if (!fin && seg != segcur) {
            const u64 x = fr[(u64)(2 * NW + seg) * 64];
            tlo = (u32)even_bits(x); thi = (u32)even_bits(x >> 1); segcur = seg;
        }
        hp <<= HALF * h; hm <<= HALF * h;                             // bit 31 - x = column c0 + x: this half's first column on top
        for (int s = 0; s < HALF + 1; ++s) {
            const u32 xp = dpp_quad_from_below(wd.pubP), xm = dpp_quad_from_below(wd.pubM);
            const int cl = s - k;
            if ((u32)cl < (u32)nc) {
                const int ci = HALF * h + cl;
                const u32 cph = k ? xp : hp << cl, cmh = k ? xm : hm << cl;
                const u32 ma = (u32)__builtin_amdgcn_sbfe((int)tlo, (u32)ci, 1u), mb = (u32)__builtin_amdgcn_sbfe((int)thi, (u32)ci, 1u);
                const u32 el = bop<0x60>(nlo_l ^ ma, nhi_l, mb), eh = bop<0x60>(nlo_h ^ ma, nhi_h, mb);
                u32 phh, mhh; u64 rA, rB;
                fast_advance<true>(wd, el, eh, cph, cmh, 0u, phh, mhh, rA, rB);
                ++words_done;
                // the walk's 64 rows of this column: low part from word w1 - 1 (written first: its lane is one step ahead), high part from word w1
                if (k == 0) { if (sh) slices[ci][q] = make_ulonglong2(rA >> sh, rB >> sh); }
                else if (!sh) slices[ci][q] = make_ulonglong2(rA, rB);
                else {
                    const u64 pa = rA << (64 - sh), pb = rB << (64 - sh);
                    if (w == 0) slices[ci][q] = make_ulonglong2(pa, pb);          // rows above the matrix are never looked at
                    else {      // (LDS atomics without a return value: nothing to wait for)
                        unsigned long long* dst = reinterpret_cast<unsigned long long*>(&slices[ci][q]);
                        __hip_atomic_fetch_or(dst, pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_or(dst + 1, pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        __syncthreads();
        // ---- the walk (walk_block of dp_core.h on the slices), every lane of the quad the same.  Two forms of the loop: once every
        // block of the wave has seen its run of 8 matches and no block keeps its ops (every block but the first one or two of an
        // extension, when only records are wanted) a step only counts columns and matches
        const bool lean = !__any(!fin && (!hit || store));
        bool out = false;                                             // walked out of the matrix: the block is done
        if (!fin) {
            if (lean) {
                for (;;) {
                    if (c < c0 || r < rb) break;                      // out of the segment / of the rows kept: the next segment (or this one again)
                    const ulonglong2 v = slices[c - c0][q];
                    const int bit = r - rb;
                    const u32 a = (u32)(v.x >> bit) & 1u, b = (u32)(v.y >> bit) & 1u;
                    const int drow = 1 - (int)(b & (a ^ 1u)), dcol = 1 - (int)(a & (b ^ 1u));
                    ++n; nmat += (int)((a | b) ^ 1u);
                    r -= drow; c -= dcol;
                    if ((r | c) < 0) { out = true; break; }
                }
            } else {
                for (;;) {
                    if (c < c0 || r < rb) break;
                    const ulonglong2 v = slices[c - c0][q];
                    const int bit = r - rb;
                    const u32 a = (u32)(v.x >> bit) & 1u, b = (u32)(v.y >> bit) & 1u;
                    const int op = (int)(a | (b << 1));
                    const int drow = 1 - (int)(b & (a ^ 1u)), dcol = 1 - (int)(a & (b ^ 1u));
                    const int mt = (int)((a | b) ^ 1u);
                    if (store && j == 0) { if (n < MAXOPS) ops[(size_t)n * 64] = (u8)op; else atomicExch(err_flag, 20); }
                    ++n; nmat += mt;
                    if (!hit) {
                        nq += drow; nt += dcol;
                        m = mt ? m + 1 : 0;
                        if (m == mlen) { hit = 1; acnt = n; qcnt = nq; tcnt = nt; mcnt = nmat; }
                    }
                    r -= drow; c -= dcol;
                    if ((r | c) < 0) { out = true; break; }
                }
            }
        }
        if (out) {
            // out of the first column: the rows left are inserts; out of the first row: the columns left are deletes
            const int kop = c < 0 ? 1 : 2, k = c < 0 ? r + 1 : c + 1;
            if (store && j == 0) for (int i = 0; i < k; ++i) { if (n + i < MAXOPS) ops[(size_t)(n + i) * 64] = (u8)kop; else atomicExch(err_flag, 20); }
            n += k;
            if (!hit && k > 0) m = 0;
            fin = true;
            if (j == 0) { WalkOut o; o.n = n; o.nmat = nmat; o.m = m; o.hit = hit; o.acnt = acnt; o.qcnt = qcnt; o.tcnt = tcnt; o.mcnt = mcnt; wout[item] = o; }
        }
        __syncthreads();             // the slices are read before the next segment overwrites them
    }
    for (int o = 32; o > 0; o >>= 1) words_done += (u32)__shfl_xor((int)words_done, o);
    if (lane == 0 && words_done) { stat_add(stats, 0, (unsigned long long)words_done); stat_add(stats, 4, (unsigned long long)words_done); }
}


// k_rcwalk2 with the walk taken off the quads.  In k_rcwalk2 the four lanes of a quad recompute together and then all four
// run the SAME walk: 16 distinct walks on 64 lanes.  Here a workgroup is four waves = 64 blocks: every wave recomputes the two-word window of
// its 16 blocks exactly as k_rcwalk2 does (quad = block, 2 words x 2 half-segments), the slices of all 64 blocks land in LDS ([column][block],
// 32 KB: the same 512 bytes per block in flight), and ONE wave walks all 64 blocks, a lane each - the walk's instructions are issued once per 64
// blocks instead of once per 16, and the three other waves wait at the barrier without taking issue slots.  Which wave walks is a hash of the
// workgroup's index, so that the walkers of the workgroups resident on a CU spread over its SIMDs.  The walker hands (r, c, done, all done) of
// every block back through the block's own column-0 slice (the quad that reads it is the only writer of that block's slices: no third barrier).
template <int NW, int TW, int COLS, int MAXOPS>
__global__ void __launch_bounds__(256)
k_rcwalk2w(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, const ulonglong2* __restrict__ ckpt,
           const u64* __restrict__ hcar, const BlockResult* __restrict__ results, const ExtTask* __restrict__ tasks, int keep_cols, int tail_match_len, u8* __restrict__ ops_pool,
           WalkOut* __restrict__ wout, unsigned long long* __restrict__ stats, int* __restrict__ err_flag, u32 epoch, u32 lo, u32 hi, u32 opts)
{
    constexpr int FW = 2 * NW + TW, SEG = kRcSeg, HALF = SEG / 2, CK = RcGeom<COLS>::kCk, SEGS = RcGeom<COLS>::kSeg;
    static_assert(COLS < 4096, "the hand-over word keeps r and c in 12 bits each");
#ifdef NECAT_RC_NOPF
    const bool pf = false;                                            // (tools/rcwalk_microbench.hip: the kernel without the prefetch's registers)
#else
    const bool pf = (opts & 1u) != 0;                                 // prefetch the next segment's inputs (NECAT_RC_PREFETCH)
#endif
    if (opts & 8u) __builtin_amdgcn_s_setprio(3);                     // (NECAT_RC_PRIO bits 1 / 4: every wave of this launch above the other streams' kernels; bits 8 / 16 -> opts 16: the walker only, below)
    __shared__ ulonglong2 slices[SEG][64];
    const ListView lv = list_view(n_host, n_dev, capA);
    const bool all = ((epoch >> 27) & 1u) != 0, ragged = ((epoch >> 26) & 1u) != 0;
    const u64 first = (u64)lo + (u64)blockIdx.x * 64, lim = (all || ragged) ? lv.n : lv.nf, end = lim < hi ? lim : hi;
    if (first >= end || (ragged && first + 64 <= (u64)lv.nf16)) return;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int q = lane >> 2, j = lane & 3, k = j & 1, h = j >> 1;
    const int rbk = 16 * wave + q;                                    // the block this lane's quad recomputes
    const u64 grp = first >> 6;                                       // the 64 work indices of the workgroup are one 64-item group
    const bool walker = wave == (int)((blockIdx.x * 0x9E3779B1u) >> 30);
    auto usable = [&](u64 item, BlockItem& it) { return item < end && !(ragged && item < (u64)lv.nf16) && list_item(lv, items, item, it); };
    // ---- the recomputing role: block rbk
    const u64 item = first + (u64)rbk;
    const u64* fr = frag + grp * FW * 64 + rbk;
    int r = 0, c = -1;
    bool fin = true;
    {
        BlockItem it0;
        if (usable(item, it0)) {
            const BlockResult br = results[item];
            if (!(br.words & kWideFlag) && br.dist >= 0) { fin = false; r = it0.qn - 1; c = br.endc; }
        }
    }
    // ---- the walking role (every wave sets it up - the loop's first test needs every block's state - only the walker's is used after that)
    const u64 witem = first + (u64)lane;
    int wr = 0, wc = -1;
    bool wfin = true, store = false;
    int mlen = kOcaMatCnt;
    {
        BlockItem it0;
        if (usable(witem, it0)) {
            const BlockResult br = results[witem];
            if (!(br.words & kWideFlag) && br.dist >= 0) {
                wfin = false; wr = it0.qn - 1; wc = br.endc;
                if (tasks) { const ExtTask& t = tasks[it0.task]; store = keep_cols || !t.found; if (t.last) mlen = tail_match_len; }
                else store = true;
            }
        }
    }
    bool all_fin = __all(wfin);
    int n = 0, nmat = 0, m = 0, hit = 0, nq = 0, nt = 0, acnt = 0, qcnt = 0, tcnt = 0, mcnt = 0;
    u8* const ops = ops_pool + (size_t)grp * MAXOPS * 64 + lane;
    int wcur = -1; u32 nlo_l = 0, nlo_h = 0, nhi_l = 0, nhi_h = 0;
    int segcur = -1; u32 tlo = 0, thi = 0;
    u32 words_done = 0;
    // What the NEXT segment will need, fetched a whole segment ahead (NECAT_RC_PREFETCH, `pf`): the walk leaves a segment through its first
    // column almost always (the next segment is seg - 1) and it moves up about as many rows as it crossed columns, so the words it will stand
    // in are this segment's pair or the pair one word up - the checkpoints / deltas / query planes of both, and the target word of seg - 1, are
    // loaded while this segment is recomputed and walked, and used if the guess was right (otherwise loaded then, as without the prefetch).
    int p_seg = -2, p_w = -100000;                                   // the prefetched segment, and the word p_ck0 belongs to (p_ck1: p_w - 1)
    u64 p_ck0x = 0, p_ck0y = 0, p_ck1x = 0, p_ck1y = 0;               // (scalars, selected: arrays indexed by `pi` / vector temporaries end up in scratch)
    u64 p_hc0 = 0, p_hc1 = 0, p_tg = 0, p_qa = 0, p_qb = 0;           // (p_hc: deltas of the word above p_ck's; p_qa / p_qb: planes of word p_w - 1)
    // (opts bits 1 / 2, timing experiments only: the walk of a segment done twice - once into a sink - / the recompute of a segment done twice: what
    // each phase costs is the difference to the plain run, profiles/NOTES_r04.md 3)
    // Compiled in only with -DNECAT_RC_TIMING (their registers cost the kernel a wave per SIMD).
#ifdef NECAT_RC_TIMING
    int sink = 0;
#endif
    while (!all_fin) {
#ifdef NECAT_RC_TIMING
        for (int rrep = (opts & 4u) ? 0 : 1; rrep < 2; ++rrep)
#endif
        {   // ---- recompute (k_rcwalk2's, with q -> rbk in the slices)
            const int seg = c >> 5, c0 = seg * SEG;
            const int rb = r - 63, sh = rb & 63, w1 = r >> 6;
            const int w = w1 - 1 + k;
            const int nc0 = c - c0 - HALF * h + 1;
            const int nc = (fin || w < 0 || nc0 < 0) ? 0 : (nc0 > HALF ? HALF : nc0);
            const bool live = nc > 0;
            const bool hit = pf && seg == p_seg && (w == p_w || w == p_w - 1);
            const int pi = w == p_w ? 0 : 1;
            if (live && w != wcur) {
                u64 a, bq;
                if (hit && pi == 1) { a = p_qa; bq = p_qb; } else { a = fr[(u64)w * 64]; bq = fr[(u64)(NW + w) * 64]; }
                nlo_l = (u32)a; nlo_h = (u32)(a >> 32); nhi_l = (u32)bq; nhi_h = (u32)(bq >> 32); wcur = w;
            }
            FastWord wd; wd.Pv = ~0ULL; wd.Mv = 0ULL; wd.pubP = 0x80000000u; wd.pubM = 0u;
            const int slot = 2 * seg + h - 1;
            if (live && slot >= 0) {
                if (hit) { wd.Pv = pi ? p_ck1x : p_ck0x; wd.Mv = pi ? p_ck1y : p_ck0y; }
                else { const ulonglong2 v = ckpt[rc_at<NW>(item - lo, CK, (size_t)slot, (size_t)w)]; wd.Pv = v.x; wd.Mv = v.y; }
            }
            u32 hp = 0xffffffffu, hm = 0u;
            if (live && k == 0 && w > 0) { const u64 v = hit ? (pi ? p_hc1 : p_hc0) : hcar[rc_at<NW>(item - lo, SEGS, (size_t)seg, (size_t)(w - 1))]; hp = (u32)v; hm = (u32)(v >> 32); }
            if (!fin && seg != segcur) {
                const u64 x = (pf && seg == p_seg) ? p_tg : fr[(u64)(2 * NW + seg) * 64];
                tlo = (u32)even_bits(x); thi = (u32)even_bits(x >> 1); segcur = seg;
            }
            if (pf && !fin && seg > 0) {
                // (issued now, consumed a segment later: the loads fly under this segment's recompute, both barriers and the walk)
                p_seg = seg - 1; p_w = w;
                const int ns = 2 * (seg - 1) + h - 1;
                const size_t ib = (size_t)(item - lo);
                if (ns >= 0 && w >= 0) { const ulonglong2 v = ckpt[rc_at<NW>(ib, CK, (size_t)ns, (size_t)w)]; p_ck0x = v.x; p_ck0y = v.y; }
                if (ns >= 0 && w >= 1) { const ulonglong2 v = ckpt[rc_at<NW>(ib, CK, (size_t)ns, (size_t)(w - 1))]; p_ck1x = v.x; p_ck1y = v.y; }
                if (k == 0 && w >= 1) p_hc0 = hcar[rc_at<NW>(ib, SEGS, (size_t)(seg - 1), (size_t)(w - 1))];
                if (k == 0 && w >= 2) p_hc1 = hcar[rc_at<NW>(ib, SEGS, (size_t)(seg - 1), (size_t)(w - 2))];
                p_tg = fr[(u64)(2 * NW + seg - 1) * 64];
                if (w >= 1) { p_qa = fr[(u64)(w - 1) * 64]; p_qb = fr[(u64)(NW + w - 1) * 64]; }
            } else p_seg = -2;
            hp <<= HALF * h; hm <<= HALF * h;
            for (int s = 0; s < HALF + 1; ++s) {
                const u32 xp = dpp_quad_from_below(wd.pubP), xm = dpp_quad_from_below(wd.pubM);
                const int cl = s - k;
                if ((u32)cl < (u32)nc) {
                    const int ci = HALF * h + cl;
                    const u32 cph = k ? xp : hp << cl, cmh = k ? xm : hm << cl;
                    const u32 ma = (u32)__builtin_amdgcn_sbfe((int)tlo, (u32)ci, 1u), mb = (u32)__builtin_amdgcn_sbfe((int)thi, (u32)ci, 1u);
                    const u32 el = bop<0x60>(nlo_l ^ ma, nhi_l, mb), eh = bop<0x60>(nlo_h ^ ma, nhi_h, mb);
                    u32 phh, mhh; u64 rA, rB;
                    fast_advance<true>(wd, el, eh, cph, cmh, 0u, phh, mhh, rA, rB);
                    ++words_done;
                    if (k == 0) { if (sh) slices[ci][rbk] = make_ulonglong2(rA >> sh, rB >> sh); }
                    else if (!sh) slices[ci][rbk] = make_ulonglong2(rA, rB);
                    else {
                        const u64 pa = rA << (64 - sh), pb = rB << (64 - sh);
                        if (w == 0) slices[ci][rbk] = make_ulonglong2(pa, pb);
                        else {
                            unsigned long long* dst = reinterpret_cast<unsigned long long*>(&slices[ci][rbk]);
                            __hip_atomic_fetch_or(dst, pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_or(dst + 1, pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (walker) {
            if (opts & 16u) __builtin_amdgcn_s_setprio(3);           // (microbenchmark: only the walking wave at raised priority, for the length of its walk)
            // ---- the walk of block `lane` (walk_block of dp_core.h on the slices): the lean form once no block of the workgroup is before its
            // run of matches or keeps its ops
            const int c0 = (wc >> 5) * SEG, rb = wr - 63;
            const bool lean = !__any(!wfin && (!hit || store));
            bool out = false;
#ifdef NECAT_RC_TIMING
            if ((opts & 2u) && !wfin) {
                int r2 = wr, c2 = wc, n2 = 0;
                for (;;) {
                    if (c2 < c0 || r2 < rb) break;
                    const ulonglong2 v = slices[c2 - c0][lane];
                    const int bit = r2 - rb;
                    const u32 a = (u32)(v.x >> bit) & 1u, b = (u32)(v.y >> bit) & 1u;
                    const int drow = 1 - (int)(b & (a ^ 1u)), dcol = 1 - (int)(a & (b ^ 1u));
                    ++n2; sink += (int)((a | b) ^ 1u);
                    r2 -= drow; c2 -= dcol;
                    if ((r2 | c2) < 0) break;
                }
                sink += n2;
            }
#endif
            if (!wfin) {
                if (lean) {
                    for (;;) {
                        if (wc < c0 || wr < rb) break;
                        const ulonglong2 v = slices[wc - c0][lane];
                        const int bit = wr - rb;
                        const u32 a = (u32)(v.x >> bit) & 1u, b = (u32)(v.y >> bit) & 1u;
                        const int drow = 1 - (int)(b & (a ^ 1u)), dcol = 1 - (int)(a & (b ^ 1u));
                        ++n; nmat += (int)((a | b) ^ 1u);
                        wr -= drow; wc -= dcol;
                        if ((wr | wc) < 0) { out = true; break; }
                    }
                } else {
                    for (;;) {
                        if (wc < c0 || wr < rb) break;
                        const ulonglong2 v = slices[wc - c0][lane];
                        const int bit = wr - rb;
                        const u32 a = (u32)(v.x >> bit) & 1u, b = (u32)(v.y >> bit) & 1u;
                        const int op = (int)(a | (b << 1));
                        const int drow = 1 - (int)(b & (a ^ 1u)), dcol = 1 - (int)(a & (b ^ 1u));
                        const int mt = (int)((a | b) ^ 1u);
                        if (store) { if (n < MAXOPS) ops[(size_t)n * 64] = (u8)op; else atomicExch(err_flag, 20); }
                        ++n; nmat += mt;
                        if (!hit) {
                            nq += drow; nt += dcol;
                            m = mt ? m + 1 : 0;
                            if (m == mlen) { hit = 1; acnt = n; qcnt = nq; tcnt = nt; mcnt = nmat; }
                        }
                        wr -= drow; wc -= dcol;
                        if ((wr | wc) < 0) { out = true; break; }
                    }
                }
            }
            if (out) {
                const int kop = wc < 0 ? 1 : 2, kk = wc < 0 ? wr + 1 : wc + 1;
                if (store) for (int i = 0; i < kk; ++i) { if (n + i < MAXOPS) ops[(size_t)(n + i) * 64] = (u8)kop; else atomicExch(err_flag, 20); }
                n += kk;
                if (!hit && kk > 0) m = 0;
                wfin = true;
                WalkOut o; o.n = n; o.nmat = nmat; o.m = m; o.hit = hit; o.acnt = acnt; o.qcnt = qcnt; o.tcnt = tcnt; o.mcnt = mcnt; wout[witem] = o;
            }
            const u32 word = wfin ? (1u << 24) : ((u32)wr | ((u32)wc << 12));
            reinterpret_cast<u32*>(&slices[0][lane])[0] = word | (__all(wfin) ? 1u << 25 : 0u);
            if (opts & 16u) __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        {
            const u32 word = reinterpret_cast<const u32*>(&slices[0][rbk])[0];
            fin = (word >> 24) & 1u; all_fin = (word >> 25) & 1u;
            r = (int)(word & 0xfffu); c = (int)((word >> 12) & 0xfffu);
        }
    }
#ifdef NECAT_RC_TIMING
    if (sink == 0x7fffffff) atomicExch(err_flag, 21);             // (keeps the sink alive; never true)
#endif
    for (int o = 32; o > 0; o >>= 1) words_done += (u32)__shfl_xor((int)words_done, o);
    if (lane == 0 && words_done) { stat_add(stats, 0, (unsigned long long)words_done); stat_add(stats, 4, (unsigned long long)words_done); }
}


}  // namespace necat
