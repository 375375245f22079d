// ext_rcwalk3.h - k_rcwalk3: the recomputing walk on a 32-DIAGONAL band (ext_bandwalk.h has the why and the per-lane cores).
// Same inputs and outputs as k_rcwalk2w (ext_rcwalk.h) - any list's work items, checkpoints / horizontal deltas from k_myers_ck / k_myers_ckg /
// k_myers_ckf in, WalkOut records and (while a task keeps them) ops out - with another division of labour:
//   workgroup  two waves = 64 work items.  Both waves recompute (32 blocks each), then ONE of them walks the 64 blocks, a lane each.
//   recompute  TWO lanes per block - the two 16-column halves of the 32-column segment - and each lane runs BOTH words of the segment's pair
//              (band_word_lo: the words of rows [r - 47, r]) one after the other: the lower word's horizontal carries reach the upper word in
//              registers (k_rcwalk2w: one lane per word, carries by DPP, the upper word's lane a step behind - 17 steps of 4 lanes where this is
//              16 steps of 2), the column's record is cut out of the four dwords of the two decision planes by one funnel shift per plane
//              (band_piece2) and stored with one ds_write_b64 (k_rcwalk2w: a ds_write_b128 and two 64-bit LDS atomics per column).  The target's
//              16 columns of a half ARE one dword of the fragment's 2-bit word: no bit-plane split per segment.
//   walk       32 column steps (band_walk_col3), the records fetched four columns at a time - their addresses do not depend on the walk.
//   LDS        16 KB + 256 B per workgroup (k_rcwalk2w: 32 KB per 64 blocks): nine workgroups per CU.
// Round 5's first version (one lane per word and half as in k_rcwalk2w, DPP hand-over of the lower word's piece) was bit-equal and no faster: 75
// vector instructions per step against 62, the kernel at ~ 0.8 of its issue bound (profiles/r05_rcwalk3_v1_microbench.txt, NOTES_r05 1).
#pragma once
#include "ext_bandwalk.h"

namespace necat {

#ifndef NECAT_RC3_WAVES
#define NECAT_RC3_WAVES 5         // waves per SIMD the register budget is cut for (16.6 KB of LDS per two waves holds 4.5; tools/rcwalk_microbench.hip builds other budgets)
#endif

// the 16 steps of a lane's half-segment: columns [0, nc) of the half (FAST: all 16, no predicate, 0 <= S < 96 throughout).  wl / wh: the pair's state
// (Pv, Mv); q*: complemented query planes of the two words; xs: the half's 16 target bases (2 bits each); hp / hm: the horizontal deltas entering the
// lower word from the word above it (bit 31 = this half's first column); S: band_piece2's row offset of the first column; dst: the half's first record
template <bool FAST>
NECAT_D u32 rc3_half(FastWord& wl, FastWord& wh, const u32 ql_nlo_l, const u32 ql_nlo_h, const u32 ql_nhi_l, const u32 ql_nhi_h,
                     const u32 qh_nlo_l, const u32 qh_nlo_h, const u32 qh_nhi_l, const u32 qh_nhi_h, u32 xs, u32 hp, u32 hm, int S, const int nc, u64* __restrict__ dst)
{
    u32 done = 0;
#pragma unroll 2
    for (int cl = 0; cl < 16; ++cl, ++S, dst += 64) {
        if (FAST || cl < nc) {
            const u32 ma = (u32)__builtin_amdgcn_sbfe((int)xs, 0u, 1u), mb = (u32)__builtin_amdgcn_sbfe((int)xs, 1u, 1u);
            xs >>= 2;
            u32 phh, mhh, phh2, mhh2; u64 a0, b0, a1, b1;
            {
                const u32 el = bop<0x60>(ql_nlo_l ^ ma, ql_nhi_l, mb), eh = bop<0x60>(ql_nlo_h ^ ma, ql_nhi_h, mb);
                fast_advance<true>(wl, el, eh, hp, hm, 0u, phh, mhh, a0, b0);
                hp <<= 1; hm <<= 1;
            }
            {
                const u32 el = bop<0x60>(qh_nlo_l ^ ma, qh_nhi_l, mb), eh = bop<0x60>(qh_nlo_h ^ ma, qh_nhi_h, mb);
                fast_advance<true>(wh, el, eh, phh, mhh, 0u, phh2, mhh2, a1, b1);
            }
            const u32 pa = band_piece2<!FAST>((u32)a0, (u32)(a0 >> 32), (u32)a1, (u32)(a1 >> 32), S);
            const u32 pb = band_piece2<!FAST>((u32)b0, (u32)(b0 >> 32), (u32)b1, (u32)(b1 >> 32), S);
            *dst = (u64)pa | ((u64)pb << 32);
            done += 2;
        }
    }
    return done;
}

template <int NW, int TW, int COLS, int MAXOPS>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(NECAT_RC3_WAVES, NECAT_RC3_WAVES)))
k_rcwalk3(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, const ulonglong2* __restrict__ ckpt,
          const u64* __restrict__ hcar, const BlockResult* __restrict__ results, const ExtTask* __restrict__ tasks, int keep_cols, int tail_match_len, u8* __restrict__ ops_pool,
          WalkOut* __restrict__ wout, unsigned long long* __restrict__ stats, int* __restrict__ err_flag, u32 epoch, u32 lo, u32 hi, u32 opts)
{
    constexpr int FW = 2 * NW + TW, SEG = kRcSeg, HALF = SEG / 2, CK = RcGeom<COLS>::kCk, SEGS = RcGeom<COLS>::kSeg, GI = RcLay<NW>::kGI;
    static_assert(COLS < 4096 && NW * 64 <= 4096, "the hand-over word keeps r and c in 12 bits each");
    static_assert(NW >= 2 && HALF == 16, "a pair of words; a half-segment is one dword of a fragment's target word");
    if (opts & 8u) __builtin_amdgcn_s_setprio(3);                     // (NECAT_RC_PRIO bits 1 / 4: every wave of the walk above the other streams' kernels)
    __shared__ u64 slices[SEG][64];
    __shared__ u32 hand[64];
    const ListView lv = list_view(n_host, n_dev, capA);
    const bool all = ((epoch >> 27) & 1u) != 0, ragged = ((epoch >> 26) & 1u) != 0;
    const u64 first = (u64)lo + (u64)blockIdx.x * 64, lim = (all || ragged) ? lv.n : lv.nf, end = lim < hi ? lim : hi;
    if (first >= end || (ragged && first + 64 <= (u64)lv.nf16)) return;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int q = lane >> 1, h = lane & 1;
    const int rbk = 32 * wave + q;                                    // the block this lane pair recomputes
    const u64 grp = first >> 6;                                       // the 64 work indices of the workgroup are one 64-item group
    const bool walker = wave == (int)((blockIdx.x * 0x9E3779B1u) >> 31);
    auto usable = [&](u64 item, BlockItem& it) { return item < end && !(ragged && item < (u64)lv.nf16) && list_item(lv, items, item, it); };
    // ---- the recomputing role: block rbk, half h
    const u64 item = first + (u64)rbk;
    const u64* const fr = frag + grp * FW * 64 + rbk;
    const ulonglong2* const ck_blk = ckpt + (size_t)(((item - lo) / GI) * (u64)(CK * NW * GI) + (item - lo) % GI);      // rc_at<NW>(item - lo, CK, 0, 0)
    const u64* const hc_blk = hcar + (size_t)(((item - lo) / GI) * (u64)(SEGS * NW * GI) + (item - lo) % GI);
    int r = 0, c = -1;
    bool fin = true;
    {
        BlockItem it0;
        if (usable(item, it0)) {
            const BlockResult br = results[item];
            if (!(br.words & kWideFlag) && br.dist >= 0) { fin = false; r = it0.qn - 1; c = br.endc; }
        }
    }
    // ---- the walking role (both waves set it up - the loop's first test needs every block's state - only the walker's is used after that)
    const u64 witem = first + (u64)lane;
    BandWalk bw; bw.r = 0; bw.c = -1; bw.p = kBandP0; bw.n = bw.nmat = 0; bw.m = bw.hit = bw.nq = bw.nt = 0; bw.acnt = bw.qcnt = bw.tcnt = bw.mcnt = 0;
    bool wfin = true, store = false;
    int mlen = kOcaMatCnt;
    {
        BlockItem it0;
        if (usable(witem, it0)) {
            const BlockResult br = results[witem];
            if (!(br.words & kWideFlag) && br.dist >= 0) {
                wfin = false; bw.r = it0.qn - 1; bw.c = br.endc;
                if (tasks) { const ExtTask& t = tasks[it0.task]; store = keep_cols || !t.found; if (t.last) mlen = tail_match_len; }
                else store = true;
            }
        }
    }
    bool all_fin = __all(wfin);
    u8* const ops = ops_pool + (size_t)grp * MAXOPS * 64 + lane;
    auto st_op = [&](int i, int op) { ops[(size_t)i * 64] = (u8)op; };
    int wcur = -1;
    u32 ql_nlo_l = 0, ql_nlo_h = 0, ql_nhi_l = 0, ql_nhi_h = 0, qh_nlo_l = 0, qh_nlo_h = 0, qh_nhi_l = 0, qh_nhi_h = 0;      // the query planes of the pair's two words
    int segcur = -1; u32 xt = 0;                                      // this half's 16 target bases of the current segment
    u32 words_done = 0;
    u64* const dst = &slices[HALF * h][rbk ^ (h << 3)];               // [column][block ^ 8 * (column / 16)]: the two halves a 16-lane store group holds hit different banks
    while (!all_fin) {
        {   // ---- recompute: both words of the pair over this lane's half of the segment's columns, the 32-diagonal record of every column
            const int seg = c >> 5, c0 = seg * SEG;
            const int wlo = band_word_lo(r), whi = wlo + 1 < NW ? wlo + 1 : NW - 1;
            const int nc0 = c - c0 - HALF * h + 1;
            const int nc = (fin || nc0 < 0) ? 0 : (nc0 > HALF ? HALF : nc0);
            const bool live = nc > 0;
            if (live && wlo != wcur) {
                const u64 a = fr[(u64)wlo * 64], bq = fr[(u64)(NW + wlo) * 64], a2 = fr[(u64)whi * 64], b2 = fr[(u64)(NW + whi) * 64];
                ql_nlo_l = (u32)a; ql_nlo_h = (u32)(a >> 32); ql_nhi_l = (u32)bq; ql_nhi_h = (u32)(bq >> 32);
                qh_nlo_l = (u32)a2; qh_nlo_h = (u32)(a2 >> 32); qh_nhi_l = (u32)b2; qh_nhi_h = (u32)(b2 >> 32); wcur = wlo;
            }
            FastWord wl, wh; wl.Pv = wh.Pv = ~0ULL; wl.Mv = wh.Mv = 0ULL; wl.pubP = wh.pubP = 0u; wl.pubM = wh.pubM = 0u;
            const int slot = 2 * seg + h - 1;                         // the state before this half's first column
            if (live && slot >= 0) {
                const ulonglong2* const p = ck_blk + (size_t)((slot * NW + wlo) * GI);
                const ulonglong2 v = p[0], v2 = p[(whi - wlo) * GI];
                wl.Pv = v.x; wl.Mv = v.y; wh.Pv = v2.x; wh.Mv = v2.y;
            }
            u32 hp = 0xffffffffu, hm = 0u;                            // word 0: the top row's boundary (+1 per column)
            if (live && wlo > 0) { const u64 v = hc_blk[(size_t)((seg * NW + wlo - 1) * GI)]; hp = (u32)v << (HALF * h); hm = (u32)(v >> 32) << (HALF * h); }
            if (!fin && seg != segcur) { const u64 x = fr[(u64)(2 * NW + seg) * 64]; xt = h ? (u32)(x >> 32) : (u32)x; segcur = seg; }
            const int S = c0 + HALF * h + (r - c) - kBandP0 - 64 * wlo;      // band_piece2's row offset of this half's first column
            // (fast form: every lane of the wave does all 16 columns of its half - or none that anybody looks at - and no record starts above its pair)
            if (__all(fin || (nc == HALF && c0 + (r - c) - kBandP0 - 64 * wlo >= 0)))
                words_done += rc3_half<true>(wl, wh, ql_nlo_l, ql_nlo_h, ql_nhi_l, ql_nhi_h, qh_nlo_l, qh_nlo_h, qh_nhi_l, qh_nhi_h, xt, hp, hm, S, HALF, dst);
            else
                words_done += rc3_half<false>(wl, wh, ql_nlo_l, ql_nlo_h, ql_nhi_l, ql_nhi_h, qh_nlo_l, qh_nlo_h, qh_nhi_l, qh_nhi_h, xt, hp, hm, S, nc, dst);
        }
        __syncthreads();
        if (walker) {
            if (opts & 16u) __builtin_amdgcn_s_setprio(3);           // (NECAT_RC_PRIO bits 8 / 16: only the walking wave, for the length of its walk)
            const int xin = bw.c & (SEG - 1);
            bw.p = kBandP0;
            bool alive = !wfin;
            int ovf = 0;
            const u64* const src = &slices[0][lane];
            if (__all(wfin || xin == SEG - 1)) {
#pragma unroll 1
                for (int x0 = SEG - 4; x0 >= 0; x0 -= 4) {
                    const int f = ((lane ^ ((x0 >> 4) << 3)) - lane);   // (the four columns of a group are in one half of the segment)
                    const u64 v3 = src[(x0 + 3) * 64 + f], v2 = src[(x0 + 2) * 64 + f], v1 = src[(x0 + 1) * 64 + f], v0 = src[x0 * 64 + f];
                    if (!__any(alive)) break;
                    band_walk_col3<MAXOPS>(bw, alive, true, (u32)v3, (u32)(v3 >> 32), mlen, store, st_op, ovf);
                    band_walk_col3<MAXOPS>(bw, alive, true, (u32)v2, (u32)(v2 >> 32), mlen, store, st_op, ovf);
                    band_walk_col3<MAXOPS>(bw, alive, true, (u32)v1, (u32)(v1 >> 32), mlen, store, st_op, ovf);
                    band_walk_col3<MAXOPS>(bw, alive, true, (u32)v0, (u32)(v0 >> 32), mlen, store, st_op, ovf);
                }
            } else {
#pragma unroll 1
                for (int x = SEG - 1; x >= 0; --x) {
                    const u64 v = src[x * 64 + ((lane ^ ((x >> 4) << 3)) - lane)];
                    if (!__any(alive)) break;
                    band_walk_col3<MAXOPS>(bw, alive, x <= xin, (u32)v, (u32)(v >> 32), mlen, store, st_op, ovf);
                }
            }
            if (ovf) atomicExch(err_flag, 20);
            if (!wfin && band_walk_why(bw, alive) == 2) {
                // out of the first column: the rows left are inserts; out of the first row: the columns left are deletes
                const int kop = bw.c < 0 ? 1 : 2, kk = bw.c < 0 ? bw.r + 1 : bw.c + 1;
                if (store) for (int i = 0; i < kk; ++i) { if (bw.n + i < MAXOPS) st_op(bw.n + i, kop); else atomicExch(err_flag, 20); }
                bw.n += kk;
                if (!bw.hit && kk > 0) bw.m = 0;
                wfin = true;
                WalkOut o; o.n = bw.n; o.nmat = bw.nmat; o.m = bw.m; o.hit = bw.hit; o.acnt = bw.acnt; o.qcnt = bw.qcnt; o.tcnt = bw.tcnt; o.mcnt = bw.mcnt; wout[witem] = o;
            }
            const u32 word = wfin ? (1u << 24) : ((u32)bw.r | ((u32)bw.c << 12));
            hand[lane] = word | (__all(wfin) ? 1u << 25 : 0u);
            if (opts & 16u) __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        {
            const u32 word = hand[rbk];
            fin = (word >> 24) & 1u; all_fin = (word >> 25) & 1u;
            r = (int)(word & 0xfffu); c = (int)((word >> 12) & 0xfffu);
        }
    }
    for (int o = 32; o > 0; o >>= 1) words_done += (u32)__shfl_xor((int)words_done, o);
    if (lane == 0 && words_done) { stat_add(stats, 0, (unsigned long long)words_done); stat_add(stats, 4, (unsigned long long)words_done); }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// k_rcwalk3p: k_rcwalk3 with the two phases of a segment SIDE BY SIDE.  In k_rcwalk2w and k_rcwalk3 a workgroup's segment is a chain - load the checkpoints,
// recompute, barrier, walk, barrier - and the wave(s) that do not walk wait: 12 us per segment for a workgroup alone (k_rcwalk2w: 9), 17 segments per block,
// which is what the small rounds of a step cost (a launch of a few thousand blocks: 170 - 200 us whatever its size) and why the big ones reach half of the
// kernel's issue rate.  Here a workgroup is three waves with fixed roles - two recompute (32 blocks each, rc3_half), one walks (64 blocks) - and two record
// buffers: while the walker is in the records of step k the recomputing waves make those of step k + 1.  What step k + 1 needs before the walk of step k is over
// is the cell it will be entered at; it is PREDICTED - the segment before this one, on the diagonal this one was entered on (a walk drifts by a few diagonals
// per segment: 32 diagonals hold the entry bit 4 .. 27 and the drift inside the segment) - and the words are those that hold the predicted band's rows (any
// entry the band holds lies inside that pair, band_word_lo's argument with the band's lowest row for r - 47).  After the barrier both sides compare the block's
// real entry with what its records assumed (rc3p_valid): if they fit the walker walks them from bit 16 + (d - d_assumed); if not (the walk left its band or its
// segment early, or drifted further than the band holds) the block sits that step out, its records are made again for the real entry - the only cost of a miss
// is that one block's one step.  Nothing is ever walked on records that do not hold the cells the walk can reach: same WalkOut records and ops as k_rcwalk2w,
// bit for bit (tools/rcwalk_microbench.hip, tests/test_gpu_parity.py).
//   LDS  2 x 16 KB of records + the entries (2 x 64 words, double-buffered: the walker writes step k's while the recomputing waves read step k - 1's) + what
//        each buffer's records assumed (2 x 64 words): 33 KB, four workgroups (12 waves) per CU - all of them busy all the time.
// assumption word: bits 0-6 segment, 7-11 last column of the segment the records hold, 12-24 the band's diagonal + 4096, bit 25 "there are records"
NECAT_D u32 rc3p_pack(int seg, int c_hi, int d) { return (u32)seg | ((u32)c_hi << 7) | ((u32)(d + 4096) << 12) | (1u << 25); }
NECAT_D int rc3p_seg(u32 a) { return (int)(a & 127u); }
NECAT_D int rc3p_chi(u32 a) { return (int)((a >> 7) & 31u); }
NECAT_D int rc3p_d(u32 a) { return (int)((a >> 12) & 8191u) - 4096; }
// do records made under assumption `a` hold what a walk entering at (r, c) can reach?  p: the entry's bit in them
NECAT_D bool rc3p_valid(u32 a, int r, int c, int& p)
{
    p = kBandP0 + (r - c) - rc3p_d(a);
    return (a >> 25) != 0u && rc3p_seg(a) == (c >> 5) && (c & 31) <= rc3p_chi(a) && p >= 4 && p <= 27;
}

template <int NW, int TW, int COLS, int MAXOPS>
__global__ void __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(3, 4)))
k_rcwalk3p(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, const ulonglong2* __restrict__ ckpt,
           const u64* __restrict__ hcar, const BlockResult* __restrict__ results, const ExtTask* __restrict__ tasks, int keep_cols, int tail_match_len, u8* __restrict__ ops_pool,
           WalkOut* __restrict__ wout, unsigned long long* __restrict__ stats, int* __restrict__ err_flag, u32 epoch, u32 lo, u32 hi, u32 opts)
{
    constexpr int FW = 2 * NW + TW, SEG = kRcSeg, HALF = SEG / 2, CK = RcGeom<COLS>::kCk, SEGS = RcGeom<COLS>::kSeg, GI = RcLay<NW>::kGI;
    static_assert(COLS < 4096 && NW * 64 <= 4096, "an entry word keeps r and c in 12 bits each, an assumption word the segment in 7");
    static_assert(NW >= 2 && HALF == 16, "a pair of words; a half-segment is one dword of a fragment's target word");
    if (opts & 8u) __builtin_amdgcn_s_setprio(3);
    __shared__ u64 rec[2][SEG][64];
    __shared__ u32 hand[2][64], assume[2][64];
    const ListView lv = list_view(n_host, n_dev, capA);
    const bool all = ((epoch >> 27) & 1u) != 0, ragged = ((epoch >> 26) & 1u) != 0;
    const u64 first = (u64)lo + (u64)blockIdx.x * 64, lim = (all || ragged) ? lv.n : lv.nf, end = lim < hi ? lim : hi;
    if (first >= end || (ragged && first + 64 <= (u64)lv.nf16)) return;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const u64 grp = first >> 6;
    auto usable = [&](u64 item, BlockItem& it) { return item < end && !(ragged && item < (u64)lv.nf16) && list_item(lv, items, item, it); };
    auto entry_of = [&](u64 item, int& r, int& c, BlockItem& it0) -> bool {        // the cell a block's walk starts at; false: nothing to walk
        if (!usable(item, it0)) return false;
        const BlockResult br = results[item];
        if ((br.words & kWideFlag) || br.dist < 0) return false;
        r = it0.qn - 1; c = br.endc;
        return true;
    };
    if (wave < 2) {
        // ================================================================ the recomputing waves: block rbk, half h
        const int q = lane >> 1, h = lane & 1;
        const int rbk = 32 * wave + q;
        const u64 item = first + (u64)rbk;
        const u64* const fr = frag + grp * FW * 64 + rbk;
        const ulonglong2* const ck_blk = ckpt + (size_t)(((item - lo) / GI) * (u64)(CK * NW * GI) + (item - lo) % GI);      // rc_at<NW>(item - lo, CK, 0, 0)
        const u64* const hc_blk = hcar + (size_t)(((item - lo) / GI) * (u64)(SEGS * NW * GI) + (item - lo) % GI);
        int r = 0, c = -1;
        bool fin;
        { BlockItem it0; fin = !entry_of(item, r, c, it0); }
        u32 a_prev = 0u;
        int wcur = -1;
        u32 ql_nlo_l = 0, ql_nlo_h = 0, ql_nhi_l = 0, ql_nhi_h = 0, qh_nlo_l = 0, qh_nlo_h = 0, qh_nhi_l = 0, qh_nhi_h = 0;
        int segcur = -1; u32 xt = 0;
        u32 words_done = 0;
        for (u32 it = 0;; ++it) {
            const u32 buf = it & 1u;
            if (it) { const u32 word = hand[buf ^ 1u][rbk]; fin = (word >> 24) & 1u; r = (int)(word & 0xfffu); c = (int)((word >> 12) & 0xfffu); }
            // what to make records for: the step after the one the walker is in now (its records fit: predicted), or the block's real entry (first step / a miss)
            u32 a = 0u;
            if (!fin) {
                int p;
                if (it && rc3p_valid(a_prev, r, c, p)) { if ((c >> 5) >= 1) a = rc3p_pack((c >> 5) - 1, SEG - 1, r - c); }
                else a = rc3p_pack(c >> 5, c & (SEG - 1), r - c);
            }
            a_prev = a;
            if (h == 0) assume[buf][rbk] = a;
            {
                const bool have = a != 0u;
                const int seg = rc3p_seg(a), c0 = seg * SEG, d = rc3p_d(a);
                const int rb0 = c0 + d - kBandP0;                           // the band's lowest row (column c0, bit 0)
                const int wlo = rb0 > 0 ? (rb0 >> 6) : 0, whi = wlo + 1 < NW ? wlo + 1 : NW - 1;
                const int nc0 = rc3p_chi(a) - HALF * h + 1;
                const int nc = (!have || nc0 < 0) ? 0 : (nc0 > HALF ? HALF : nc0);
                const bool live = nc > 0;
                if (live && wlo != wcur) {
                    const u64 x0 = fr[(u64)wlo * 64], x1 = fr[(u64)(NW + wlo) * 64], x2 = fr[(u64)whi * 64], x3 = fr[(u64)(NW + whi) * 64];
                    ql_nlo_l = (u32)x0; ql_nlo_h = (u32)(x0 >> 32); ql_nhi_l = (u32)x1; ql_nhi_h = (u32)(x1 >> 32);
                    qh_nlo_l = (u32)x2; qh_nlo_h = (u32)(x2 >> 32); qh_nhi_l = (u32)x3; qh_nhi_h = (u32)(x3 >> 32); wcur = wlo;
                }
                FastWord wl, wh; wl.Pv = wh.Pv = ~0ULL; wl.Mv = wh.Mv = 0ULL; wl.pubP = wh.pubP = 0u; wl.pubM = wh.pubM = 0u;
                const int slot = 2 * seg + h - 1;
                if (live && slot >= 0) {
                    const ulonglong2* const pk = ck_blk + (size_t)((slot * NW + wlo) * GI);
                    const ulonglong2 v = pk[0], v2 = pk[(whi - wlo) * GI];
                    wl.Pv = v.x; wl.Mv = v.y; wh.Pv = v2.x; wh.Mv = v2.y;
                }
                u32 hp = 0xffffffffu, hm = 0u;
                if (live && wlo > 0) { const u64 v = hc_blk[(size_t)((seg * NW + wlo - 1) * GI)]; hp = (u32)v << (HALF * h); hm = (u32)(v >> 32) << (HALF * h); }
                if (live && seg != segcur) { const u64 x = fr[(u64)(2 * NW + seg) * 64]; xt = h ? (u32)(x >> 32) : (u32)x; segcur = seg; }
                u64* const dst = &rec[buf][HALF * h][rbk ^ (h << 3)];
                const int S = rb0 + HALF * h - 64 * wlo;
                if (__all(!have || (nc == HALF && rb0 >= 0)))
                    { if (__any(have)) words_done += rc3_half<true>(wl, wh, ql_nlo_l, ql_nlo_h, ql_nhi_l, ql_nhi_h, qh_nlo_l, qh_nlo_h, qh_nhi_l, qh_nhi_h, xt, hp, hm, S, HALF, dst); }
                else
                    words_done += rc3_half<false>(wl, wh, ql_nlo_l, ql_nlo_h, ql_nhi_l, ql_nhi_h, qh_nlo_l, qh_nlo_h, qh_nhi_l, qh_nhi_h, xt, hp, hm, S, nc, dst);
            }
            __syncthreads();
            if ((hand[buf][0] >> 25) & 1u) break;
        }
        for (int o = 32; o > 0; o >>= 1) words_done += (u32)__shfl_xor((int)words_done, o);
        if (lane == 0 && words_done) { stat_add(stats, 0, (unsigned long long)words_done); stat_add(stats, 4, (unsigned long long)words_done); }
    } else {
        // ================================================================ the walker: block `lane`
        const u64 witem = first + (u64)lane;
        BandWalk bw; bw.r = 0; bw.c = -1; bw.p = kBandP0; bw.n = bw.nmat = 0; bw.m = bw.hit = bw.nq = bw.nt = 0; bw.acnt = bw.qcnt = bw.tcnt = bw.mcnt = 0;
        bool wfin, store = false;
        int mlen = kOcaMatCnt;
        {
            BlockItem it0;
            wfin = !entry_of(witem, bw.r, bw.c, it0);
            if (!wfin) {
                if (tasks) { const ExtTask& t = tasks[it0.task]; store = keep_cols || !t.found; if (t.last) mlen = tail_match_len; }
                else store = true;
            }
        }
        u8* const ops = ops_pool + (size_t)grp * MAXOPS * 64 + lane;
        auto st_op = [&](int i, int op) { ops[(size_t)i * 64] = (u8)op; };
        for (u32 it = 0;; ++it) {
            const u32 buf = it & 1u;
            if (it) {
                // the records of the step before (rec[buf ^ 1]): walked by the blocks whose entry they fit
                int p = 0;
                const bool go = !wfin && rc3p_valid(assume[buf ^ 1u][lane], bw.r, bw.c, p);
                if (opts & 16u) __builtin_amdgcn_s_setprio(3);
                const int xin = bw.c & (SEG - 1);
                bw.p = p;
                bool alive = go;
                int ovf = 0;
                const u64* const src = &rec[buf ^ 1u][0][lane];
                if (__all(!go || xin == SEG - 1)) {
#pragma unroll 1
                    for (int x0 = SEG - 4; x0 >= 0; x0 -= 4) {
                        const int f = ((lane ^ ((x0 >> 4) << 3)) - lane);
                        const u64 v3 = src[(x0 + 3) * 64 + f], v2 = src[(x0 + 2) * 64 + f], v1 = src[(x0 + 1) * 64 + f], v0 = src[x0 * 64 + f];
                        if (!__any(alive)) break;
                        band_walk_col3<MAXOPS>(bw, alive, true, (u32)v3, (u32)(v3 >> 32), mlen, store, st_op, ovf);
                        band_walk_col3<MAXOPS>(bw, alive, true, (u32)v2, (u32)(v2 >> 32), mlen, store, st_op, ovf);
                        band_walk_col3<MAXOPS>(bw, alive, true, (u32)v1, (u32)(v1 >> 32), mlen, store, st_op, ovf);
                        band_walk_col3<MAXOPS>(bw, alive, true, (u32)v0, (u32)(v0 >> 32), mlen, store, st_op, ovf);
                    }
                } else {
#pragma unroll 1
                    for (int x = SEG - 1; x >= 0; --x) {
                        const u64 v = src[x * 64 + ((lane ^ ((x >> 4) << 3)) - lane)];
                        if (!__any(alive)) break;
                        band_walk_col3<MAXOPS>(bw, alive, x <= xin, (u32)v, (u32)(v >> 32), mlen, store, st_op, ovf);
                    }
                }
                if (ovf) atomicExch(err_flag, 20);
                if (go && band_walk_why(bw, alive) == 2) {
                    // out of the first column: the rows left are inserts; out of the first row: the columns left are deletes
                    const int kop = bw.c < 0 ? 1 : 2, kk = bw.c < 0 ? bw.r + 1 : bw.c + 1;
                    if (store) for (int i = 0; i < kk; ++i) { if (bw.n + i < MAXOPS) st_op(bw.n + i, kop); else atomicExch(err_flag, 20); }
                    bw.n += kk;
                    if (!bw.hit && kk > 0) bw.m = 0;
                    wfin = true;
                    WalkOut o; o.n = bw.n; o.nmat = bw.nmat; o.m = bw.m; o.hit = bw.hit; o.acnt = bw.acnt; o.qcnt = bw.qcnt; o.tcnt = bw.tcnt; o.mcnt = bw.mcnt; wout[witem] = o;
                }
                if (opts & 16u) __builtin_amdgcn_s_setprio(0);
            }
            const bool done = __all(wfin);
            hand[buf][lane] = (wfin ? (1u << 24) : ((u32)bw.r | ((u32)bw.c << 12))) | (done ? 1u << 25 : 0u);
            __syncthreads();
            if (done) break;
        }
    }
}

}  // namespace necat
