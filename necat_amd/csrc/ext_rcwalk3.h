// ext_rcwalk3.h - k_rcwalk3: the recomputing walk on a 32-DIAGONAL band (ext_bandwalk.h has the why and the per-lane cores).
// Same inputs, outputs and launch geometry as k_rcwalk2w (ext_rcwalk.h): a workgroup of four waves takes 64 work items of any list; every wave
// recomputes, quad by quad, the two words its 16 blocks' walks stand in (2 words x 2 half-segments of 16 columns from the checkpoints and the
// horizontal deltas k_myers_ck / k_myers_ckg / k_myers_ckf left), one wave walks the 64 blocks, a lane each.  What changed:
//   recompute  the lane of the lower word (k = 0, one step ahead) publishes its piece of the column's 32-diagonal record in two registers, the lane
//              of the upper word reads them by DPP, ORs its own piece in and stores 8 bytes: one ds_write_b64 per column (was: ds_write_b128 + two
//              64-bit LDS atomics); [column][block ^ 8 * (column / 16)]: the two half-segments a 16-lane store group holds hit different banks;
//   walk       32 column steps (band_walk_col), the records fetched four columns at a time - their addresses do not depend on the walk;
//   LDS        16 KB + 256 B per workgroup instead of 32 KB; hand-over of (r, c, done) through 64 words of their own.
#pragma once
#include "ext_bandwalk.h"

namespace necat {

#ifndef NECAT_RC3_WAVES
#define NECAT_RC3_WAVES 8         // waves per SIMD the register budget is cut for (tools/rcwalk_microbench.hip builds 6 and 7 as well)
#endif

// band_piece for the device: the dword pair that holds bits [rel, rel + 32) chosen by rel >> 5, one funnel shift
NECAT_D u32 band_piece_dev(const u32 lo, const u32 hi, const int q, const u32 sh)
{
    const u32 x1 = q == 0 ? hi : (q == -1 ? lo : 0u), x0 = q == 0 ? lo : (q == 1 ? hi : 0u);
    return __builtin_amdgcn_alignbit(x1, x0, sh);
}

template <int NW, int TW, int COLS, int MAXOPS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NECAT_RC3_WAVES, NECAT_RC3_WAVES)))
k_rcwalk3(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, const ulonglong2* __restrict__ ckpt,
          const u64* __restrict__ hcar, const BlockResult* __restrict__ results, const ExtTask* __restrict__ tasks, int keep_cols, int tail_match_len, u8* __restrict__ ops_pool,
          WalkOut* __restrict__ wout, unsigned long long* __restrict__ stats, int* __restrict__ err_flag, u32 epoch, u32 lo, u32 hi, u32 opts)
{
    constexpr int FW = 2 * NW + TW, SEG = kRcSeg, HALF = SEG / 2, CK = RcGeom<COLS>::kCk, SEGS = RcGeom<COLS>::kSeg;
    static_assert(COLS < 4096 && NW * 64 <= 4096, "the hand-over word keeps r and c in 12 bits each");
    if (opts & 8u) __builtin_amdgcn_s_setprio(3);                     // (NECAT_RC_PRIO bits 1 / 4: every wave of the walk above the other streams' kernels)
    __shared__ u64 slices[SEG][64];
    __shared__ u32 hand[64];
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
    auto put = [&](int i, int op) { if (i < MAXOPS) ops[(size_t)i * 64] = (u8)op; else atomicExch(err_flag, 20); };
    int wcur = -1; u32 nlo_l = 0, nlo_h = 0, nhi_l = 0, nhi_h = 0;
    int segcur = -1; u32 tlo = 0, thi = 0;
    u32 words_done = 0;
    const int swz = h << 3;                                           // the store's block index is flipped by 8 in the segment's second half
    while (!all_fin) {
        {   // ---- recompute: this lane's word (w1 - 1 + k) over its half (h) of the segment's columns, the 32-diagonal record of every column
            const int seg = c >> 5, c0 = seg * SEG;
            const int w1 = r >> 6, w = w1 - 1 + k;
            const int nc0 = c - c0 - HALF * h + 1;
            const int nc = (fin || w < 0 || nc0 < 0) ? 0 : (nc0 > HALF ? HALF : nc0);
            const bool live = nc > 0;
            if (live && w != wcur) {
                const u64 a = fr[(u64)w * 64], bq = fr[(u64)(NW + w) * 64];
                nlo_l = (u32)a; nlo_h = (u32)(a >> 32); nhi_l = (u32)bq; nhi_h = (u32)(bq >> 32); wcur = w;
            }
            FastWord wd; wd.Pv = ~0ULL; wd.Mv = 0ULL; wd.pubP = 0x80000000u; wd.pubM = 0u;
            const int slot = 2 * seg + h - 1;
            if (live && slot >= 0) { const ulonglong2 v = ckpt[rc_at<NW>(item - lo, CK, (size_t)slot, (size_t)w)]; wd.Pv = v.x; wd.Mv = v.y; }
            u32 hp = 0xffffffffu, hm = 0u;
            if (live && k == 0 && w > 0) { const u64 v = hcar[rc_at<NW>(item - lo, SEGS, (size_t)seg, (size_t)(w - 1))]; hp = (u32)v; hm = (u32)(v >> 32); }
            if (!fin && seg != segcur) {
                const u64 x = fr[(u64)(2 * NW + seg) * 64];
                tlo = (u32)even_bits(x); thi = (u32)even_bits(x >> 1); segcur = seg;
            }
            hp <<= HALF * h; hm <<= HALF * h;
            // bit 0 of column (c0 + HALF h + cl)'s record is row (column + d0 - kBandP0), d0 = r - c; relative to this lane's word, at step s = cl + k:
            int rel = c0 + HALF * h - k + (r - c) - kBandP0 - 64 * w;
            u32 pubA = 0u, pubB = 0u;                                 // the lower word's piece of its last column (the upper word's lane reads it a step later)
            u64* const dst = &slices[HALF * h][rbk ^ swz];
            for (int s = 0; s < HALF + 1; ++s, ++rel) {
                const u32 xp = dpp_quad_from_below(wd.pubP), xm = dpp_quad_from_below(wd.pubM);
                const u32 xa = dpp_quad_from_below(pubA), xb = dpp_quad_from_below(pubB);
                const int cl = s - k;
                if ((u32)cl < (u32)nc) {
                    const int ci = HALF * h + cl;
                    const u32 cph = k ? xp : hp << cl, cmh = k ? xm : hm << cl;
                    const u32 ma = (u32)__builtin_amdgcn_sbfe((int)tlo, (u32)ci, 1u), mb = (u32)__builtin_amdgcn_sbfe((int)thi, (u32)ci, 1u);
                    const u32 el = bop<0x60>(nlo_l ^ ma, nhi_l, mb), eh = bop<0x60>(nlo_h ^ ma, nhi_h, mb);
                    u32 phh, mhh; u64 rA, rB;
                    fast_advance<true>(wd, el, eh, cph, cmh, 0u, phh, mhh, rA, rB);
                    ++words_done;
                    const int qq = rel >> 5; const u32 sh = (u32)rel & 31u;
                    const u32 pa = band_piece_dev((u32)rA, (u32)(rA >> 32), qq, sh), pb = band_piece_dev((u32)rB, (u32)(rB >> 32), qq, sh);
                    if (k == 0) { pubA = pa; pubB = pb; }
                    else dst[(size_t)cl * 64] = (u64)(pa | xa) | ((u64)(pb | xb) << 32);
                }
            }
        }
        __syncthreads();
        if (walker) {
            if (opts & 16u) __builtin_amdgcn_s_setprio(3);           // (NECAT_RC_PRIO bits 8 / 16: only the walking wave, for the length of its walk)
            const int c0 = (bw.c >> 5) * SEG, xin = bw.c - c0;
            bw.p = kBandP0;
            int st = wfin ? 3 : 0, ovf = 0;                          // 0: walking; 1: out of the band; 2: out of the matrix; 3: was done before
            auto st_op = [&](int i, int op) { ops[(size_t)i * 64] = (u8)op; };
            const u64* const src = &slices[0][lane];
#pragma unroll 1
            for (int x0 = SEG - 4; x0 >= 0; x0 -= 4) {
                const int f = ((lane ^ ((x0 >> 4) << 3)) - lane);   // (the four columns of a group are in one half of the segment)
                const u64 v3 = src[(x0 + 3) * 64 + f], v2 = src[(x0 + 2) * 64 + f], v1 = src[(x0 + 1) * 64 + f], v0 = src[x0 * 64 + f];
                if (!__any(st == 0)) break;
                band_walk_col2<MAXOPS>(bw, st, st == 0 && x0 + 3 <= xin, (u32)v3, (u32)(v3 >> 32), mlen, store, st_op, ovf);
                band_walk_col2<MAXOPS>(bw, st, st == 0 && x0 + 2 <= xin, (u32)v2, (u32)(v2 >> 32), mlen, store, st_op, ovf);
                band_walk_col2<MAXOPS>(bw, st, st == 0 && x0 + 1 <= xin, (u32)v1, (u32)(v1 >> 32), mlen, store, st_op, ovf);
                band_walk_col2<MAXOPS>(bw, st, st == 0 && x0 <= xin, (u32)v0, (u32)(v0 >> 32), mlen, store, st_op, ovf);
            }
            if (ovf) atomicExch(err_flag, 20);
            if (st == 2) {
                // out of the first column: the rows left are inserts; out of the first row: the columns left are deletes
                const int kop = bw.c < 0 ? 1 : 2, kk = bw.c < 0 ? bw.r + 1 : bw.c + 1;
                if (store) for (int i = 0; i < kk; ++i) put(bw.n + i, kop);
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

}  // namespace necat
