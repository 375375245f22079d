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
#include <type_traits>
#include "ext_bandwalk.h"

namespace necat {

#ifndef NECAT_RC3_WAVES
#define NECAT_RC3_WAVES 5         // waves per SIMD the register budget is cut for, 32-diagonal records (16.6 KB of LDS per two waves holds 4.5; tools/rcwalk_microbench.hip builds other budgets)
#endif
#ifndef NECAT_RC3_WAVES16
#define NECAT_RC3_WAVES16 7       // .. 16-diagonal records (10.7 KB per two waves holds 7.5): 72 registers
#endif

// the 16 steps of a lane's half-segment: columns [0, nc) of the half (FAST: all 16, no predicate, 0 <= S < 96 throughout).  wl / wh: the pair's state
// (Pv, Mv); q*: complemented query planes of the two words; xs: the half's 16 target bases (2 bits each); hp / hm: the horizontal deltas entering the
// lower word from the word above it (bit 31 = this half's first column); S: band_piece2's row offset of the first column; dst: the half's first record
template <bool FAST, int BW, class Rec>
NECAT_D u32 rc3_half(FastWord& wl, FastWord& wh, const u32 ql_nlo_l, const u32 ql_nlo_h, const u32 ql_nhi_l, const u32 ql_nhi_h,
                     const u32 qh_nlo_l, const u32 qh_nlo_h, const u32 qh_nhi_l, const u32 qh_nhi_h, u32 xs, u32 hp, u32 hm, int S, const int nc, Rec* __restrict__ dst)
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
            if (BW == 32) *dst = (Rec)((u64)pa | ((u64)pb << 32));
            else *dst = (Rec)((pa & 0xffffu) | (pb << 16));
            done += 2;
        }
    }
    return done;
}

// BW: diagonals per record - 32 (8 bytes per column: 16.6 KB of LDS per workgroup, 4.5 waves per SIMD) or 16 (4 bytes: 8.3 KB + the walker's state, 7 waves per
// SIMD at 72 registers; a walk leaves 16 diagonals a little more often - 3 % of the segments redone at 22 % divergence against 0.6 %, tests/host_core/check_bandwalk.cpp)
template <int NW, int TW, int COLS, int MAXOPS, int BW>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(BW == 16 ? NECAT_RC3_WAVES16 : NECAT_RC3_WAVES, BW == 16 ? NECAT_RC3_WAVES16 : NECAT_RC3_WAVES)))
k_rcwalk3(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, const ulonglong2* __restrict__ ckpt,
          const u64* __restrict__ hcar, const BlockResult* __restrict__ results, const ExtTask* __restrict__ tasks, int keep_cols, int tail_match_len, u8* __restrict__ ops_pool,
          WalkOut* __restrict__ wout, unsigned long long* __restrict__ stats, int* __restrict__ err_flag, u32 epoch, u32 lo, u32 hi, u32 opts)
{
    constexpr int FW = 2 * NW + TW, SEG = kRcSeg, HALF = SEG / 2, CK = RcGeom<COLS>::kCk, SEGS = RcGeom<COLS>::kSeg, GI = RcLay<NW>::kGI, P0 = BW / 2;
    static_assert(BW == 32 || BW == 16, "a record is two planes of 32 or of 16 diagonals");
    static_assert(COLS < 4096 && NW * 64 <= 4096, "the hand-over word keeps r and c in 12 bits each");
    static_assert(NW >= 2 && HALF == 16, "a pair of words; a half-segment is one dword of a fragment's target word");
    typedef typename std::conditional<BW == 32, u64, u32>::type Rec;
    if (opts & 8u) __builtin_amdgcn_s_setprio(3);                     // (NECAT_RC_PRIO bits 1 / 4: every wave of the walk above the other streams' kernels)
    __shared__ Rec slices[SEG][64];
    __shared__ u32 hand[64];
    // the walker's state between two segments: only one of the two waves walks, but registers are the kernel's - held in registers the 12 values would be live
    // across the recompute loop of both waves (k_rcwalk3's first form: 96 registers)
    __shared__ int wst[9][64];
    const ListView lv = list_view(n_host, n_dev, capA);
    const bool all = ((epoch >> 27) & 1u) != 0, ragged = ((epoch >> 26) & 1u) != 0;
    const u64 first = (u64)lo + (u64)blockIdx.x * 64, lim = (all || ragged) ? lv.n : lv.nf, end = lim < hi ? lim : hi;
    if (first >= end || (ragged && first + 64 <= (u64)lv.nf16)) return;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int q = lane >> 1, h = lane & 1;
    const int rbk = 32 * wave + q;                                    // the block this lane pair recomputes
    const u64 grp = first >> 6;                                       // the 64 work indices of the workgroup are one 64-item group
    const bool walker = wave == (int)((blockIdx.x * 0x9E3779B1u) >> 31);
    // ---- the walking role: block `lane`.  Flags of the block's walk: bit 0 keeps its ops, bits 8.. the run of matches its tail scan looks for
    const u64 witem = first + (u64)lane;
    u32 wflags = 0;
    bool wfin = true;
    if (walker) {
        BlockItem it0;
        int r0 = 0, c0 = 0;
        if (witem < end && !(ragged && witem < (u64)lv.nf16) && list_item(lv, items, witem, it0)) {
            const BlockResult br = results[witem];
            if (!(br.words & kWideFlag) && br.dist >= 0) {
                wfin = false; r0 = it0.qn - 1; c0 = br.endc;
                int mlen = kOcaMatCnt; bool store = true;
                if (tasks) { const ExtTask& t = tasks[it0.task]; store = keep_cols || !t.found; if (t.last) mlen = tail_match_len; }
                wflags = (store ? 1u : 0u) | ((u32)mlen << 8);
            }
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) wst[i][lane] = 0;
        hand[lane] = (wfin ? (1u << 24) : ((u32)r0 | ((u32)c0 << 12))) | (__all(wfin) ? 1u << 25 : 0u);
    }
    __syncthreads();
    // ---- the recomputing role: block rbk, half h
    const u64 item = first + (u64)rbk;
    const u64* const fr = frag + grp * FW * 64 + rbk;
    const ulonglong2* const ck_blk = ckpt + (size_t)(((item - lo) / GI) * (u64)(CK * NW * GI) + (item - lo) % GI);      // rc_at<NW>(item - lo, CK, 0, 0)
    const u64* const hc_blk = hcar + (size_t)(((item - lo) / GI) * (u64)(SEGS * NW * GI) + (item - lo) % GI);
    u8* const ops = ops_pool + (size_t)grp * MAXOPS * 64 + lane;
    auto st_op = [&](int i, int op) { ops[(size_t)i * 64] = (u8)op; };
    int wcur = -1;
    u32 ql_nlo_l = 0, ql_nlo_h = 0, ql_nhi_l = 0, ql_nhi_h = 0, qh_nlo_l = 0, qh_nlo_h = 0, qh_nhi_l = 0, qh_nhi_h = 0;      // the query planes of the pair's two words
    int segcur = -1; u32 xt = 0;                                      // this half's 16 target bases of the current segment
    u32 words_done = 0;
    Rec* const dst = &slices[HALF * h][rbk ^ (h << 3)];               // [column][block ^ 8 * (column / 16)]: the two halves a 16-lane store group holds hit different banks
    for (;;) {
        const u32 word = hand[rbk];
        if ((word >> 25) & 1u) break;
        {   // ---- recompute: both words of the pair over this lane's half of the segment's columns, the record of every column
            const bool fin = (word >> 24) & 1u;
            const int r = (int)(word & 0xfffu), c = (int)((word >> 12) & 0xfffu);
            const int seg = c >> 5, c0 = seg * SEG;
            const int wlo = band_word_lo<BW>(r), whi = wlo + 1 < NW ? wlo + 1 : NW - 1;
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
            const int S = c0 + HALF * h + (r - c) - P0 - 64 * wlo;    // band_piece2's row offset of this half's first column
            // (fast form: every lane of the wave does all 16 columns of its half - or none that anybody looks at - and no record starts above its pair)
            if (__all(fin || (nc == HALF && c0 + (r - c) - P0 - 64 * wlo >= 0)))
                words_done += rc3_half<true, BW>(wl, wh, ql_nlo_l, ql_nlo_h, ql_nhi_l, ql_nhi_h, qh_nlo_l, qh_nlo_h, qh_nhi_l, qh_nhi_h, xt, hp, hm, S, HALF, dst);
            else
                words_done += rc3_half<false, BW>(wl, wh, ql_nlo_l, ql_nlo_h, ql_nhi_l, ql_nhi_h, qh_nlo_l, qh_nlo_h, qh_nhi_l, qh_nhi_h, xt, hp, hm, S, nc, dst);
        }
        __syncthreads();
        if (walker) {
            if (opts & 16u) __builtin_amdgcn_s_setprio(3);           // (NECAT_RC_PRIO bits 8 / 16: only the walking wave, for the length of its walk)
            const u32 wword = hand[lane];
            BandWalk bw;
            bw.r = (int)(wword & 0xfffu); bw.c = (int)((wword >> 12) & 0xfffu); bw.p = P0;
            bw.n = wst[0][lane]; bw.nmat = wst[1][lane];
            { const int mh = wst[2][lane]; bw.m = mh & 0xffff; bw.hit = mh >> 16; }
            bw.nq = wst[3][lane]; bw.nt = wst[4][lane]; bw.acnt = wst[5][lane]; bw.qcnt = wst[6][lane]; bw.tcnt = wst[7][lane]; bw.mcnt = wst[8][lane];
            const bool store = (wflags & 1u) != 0;
            const int mlen = (int)(wflags >> 8);
            const int xin = bw.c & (SEG - 1);
            bool alive = !wfin;
            int ovf = 0;
            const Rec* const src = &slices[0][lane];
            auto planes = [](Rec v, u32& A, u32& B) { if (BW == 32) { A = (u32)v; B = (u32)((u64)v >> 32); } else { A = (u32)v & 0xffffu; B = (u32)v >> 16; } };
            if (__all(wfin || xin == SEG - 1)) {
#pragma unroll 1
                for (int x0 = SEG - 4; x0 >= 0; x0 -= 4) {
                    const int f = ((lane ^ ((x0 >> 4) << 3)) - lane);   // (the four columns of a group are in one half of the segment)
                    const Rec v3 = src[(x0 + 3) * 64 + f], v2 = src[(x0 + 2) * 64 + f], v1 = src[(x0 + 1) * 64 + f], v0 = src[x0 * 64 + f];
                    if (!__any(alive)) break;
                    u32 A, B;
                    planes(v3, A, B); band_walk_col3<MAXOPS, BW>(bw, alive, true, A, B, mlen, store, st_op, ovf);
                    planes(v2, A, B); band_walk_col3<MAXOPS, BW>(bw, alive, true, A, B, mlen, store, st_op, ovf);
                    planes(v1, A, B); band_walk_col3<MAXOPS, BW>(bw, alive, true, A, B, mlen, store, st_op, ovf);
                    planes(v0, A, B); band_walk_col3<MAXOPS, BW>(bw, alive, true, A, B, mlen, store, st_op, ovf);
                }
            } else {
#pragma unroll 1
                for (int x = SEG - 1; x >= 0; --x) {
                    const Rec v = src[x * 64 + ((lane ^ ((x >> 4) << 3)) - lane)];
                    if (!__any(alive)) break;
                    u32 A, B;
                    planes(v, A, B); band_walk_col3<MAXOPS, BW>(bw, alive, x <= xin, A, B, mlen, store, st_op, ovf);
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
            if (!wfin) {
                wst[0][lane] = bw.n; wst[1][lane] = bw.nmat; wst[2][lane] = bw.m | (bw.hit << 16);
                if (__any(!bw.hit)) { wst[3][lane] = bw.nq; wst[4][lane] = bw.nt; }
                wst[5][lane] = bw.acnt; wst[6][lane] = bw.qcnt; wst[7][lane] = bw.tcnt; wst[8][lane] = bw.mcnt;
            }
            hand[lane] = (wfin ? (1u << 24) : ((u32)bw.r | ((u32)bw.c << 12))) | (__all(wfin) ? 1u << 25 : 0u);
            if (opts & 16u) __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) words_done += (u32)__shfl_xor((int)words_done, o);
    if (lane == 0 && words_done) { stat_add(stats, 0, (unsigned long long)words_done); stat_add(stats, 4, (unsigned long long)words_done); }
}

}  // namespace necat
