// ext_bandwalk.h - the recomputing walk of ext_rcwalk.h rebuilt around what bounded it: LDS bytes and LDS
// instructions per block in flight, and the dependent LDS round trip of every walk step.
//
// k_rcwalk2w kept, per block and 32-column segment, the walk's decisions of the 64 ROWS [r - 63, r] - 16 bytes per column, 512 bytes
// per block in flight, five workgroups per CU - written with one ds_write_b128 (13 LDS cycles per wave-instruction) and two 64-bit LDS
// atomics per column and read back one dependent ds_read_b128 per walk step (~ 36 per segment, each: LDS latency -> two 64-bit shifts
// -> new position -> next address).  Here the decisions are kept by DIAGONAL:
//   * the walk enters a segment at (r, c) on diagonal d0 = r - c; a move keeps the diagonal (match / mismatch), lowers it by one (up:
//     a query base against a gap) or raises it by one (left).  Bit p of a column's record is the cell of that column on diagonal
//     d0 - 16 + p: 32 diagonals, the entry on bit 16 - 8 bytes per column, 256 bytes per block in flight, 16 KB per workgroup of 64
//     blocks (eight workgroups per CU), one ds_write_b64 per column;
//   * the rows a segment can touch, [r - 47, r], still lie in the two words w1 = r / 64 and w1 - 1 that the quad recomputes from the
//     checkpoints (exactly as k_rcwalk2w does: no band argument - a row never depends on a row below it); the lane of word w1 - 1
//     hands its piece of a column's record to the lane of word w1 by DPP (it is one step ahead), which ORs and stores: no LDS atomics;
//   * the walker's record address depends on the COLUMN only, so its 32 loads do not wait for the walk; the "up" moves of a column are
//     one count-leading-zeros on the column's up mask, and what is left of a column is one move to the column before it: 32 straight
//     column steps per segment instead of ~ 36 data-dependent cell steps.
// A walk that drifts out of its 32 diagonals before the segment's first column (16 net ups or 16 net lefts within 32 columns) stops
// where it is and the segment is redone from there with the band re-centred - the same "redo" k_rcwalk2w had for its 64 rows.
// The cores are NECAT_HD: tests/host_core/check_bandwalk.cpp replays them against walk_block (dp_core.h) on the CPU.
#pragma once
#include "dev_common.h"

#ifndef NECAT_ANY
#if defined(__HIP_DEVICE_COMPILE__)
#define NECAT_ANY(cond) (__ballot(cond) != 0ULL)
#else
#define NECAT_ANY(cond) (cond)
#endif
#endif

namespace necat {

constexpr int kBandP0 = 16;                      // the bit of the entry cell's diagonal in a column's record

NECAT_HD int clz32(u32 x)                        // 32 for x == 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz((int)x);
#else
    return x ? __builtin_clz(x) : 32;
#endif
}

// bits [rel, rel + 32) of a word's 64-row plane (bit i = row 64 w + i), zeros where the word has no row
NECAT_HD u32 band_piece(u64 plane, int rel)
{
    if (rel >= 64 || rel <= -32) return 0u;
    return rel >= 0 ? (u32)(plane >> rel) : ((u32)plane << (-rel));
}

// Which two words a segment entered at row r is recomputed on: the rows it can touch are [r - 47, r] (32 columns, 32 diagonals, the entry on bit 16), so the
// pair starts at the word of row r - 47 (word 0 when r < 47: the rows above the matrix then read as zero, band_piece2's general form)
// (BW diagonals, the entry on bit BW / 2: 32 columns back the band's lowest row is r - 31 - BW / 2)
template <int BW = 32> NECAT_HD int band_word_lo(int r) { return r >= 31 + BW / 2 ? (r - 31 - BW / 2) >> 6 : 0; }

// a column's record from the decision planes of the pair (lo: word wl, hi: word wl + 1): bits [S, S + 32) of the 128 rows that start at row 64 wl.
// GENERAL = false: 0 <= S < 96 (every segment entered at r >= 47); true: any S in (-64, 128), rows outside the pair read as zero.
template <bool GENERAL>
NECAT_HD u32 band_piece2(const u32 d0, const u32 d1, const u32 d2, const u32 d3, const int S)
{
    const int i = S >> 5;
    const u32 sh = (u32)S & 31u;
    u32 x0, x1;
    if (GENERAL) {
        x0 = i == 0 ? d0 : (i == 1 ? d1 : (i == 2 ? d2 : (i == 3 ? d3 : 0u)));
        x1 = i == -1 ? d0 : (i == 0 ? d1 : (i == 1 ? d2 : (i == 2 ? d3 : 0u)));
    } else {
        x0 = i == 0 ? d0 : (i == 1 ? d1 : d2);
        x1 = i == 0 ? d1 : (i == 1 ? d2 : d3);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(x1, x0, sh);
#else
    return (u32)((((u64)x1 << 32) | x0) >> sh);
#endif
}

// the walker's state of one block (walk_block's locals, dp_core.h)
struct BandWalk {
    int r, c;                    // the cell the walk stands on
    int p;                       // its bit in the current band (kBandP0 when a segment is entered)
    int n, nmat;                 // ops so far, matches among them
    int m, hit, nq, nt;          // TailScan: current run of matches, run of `mlen` seen, bases consumed while it was not
    int acnt, qcnt, tcnt, mcnt;  // .. the totals when it was
};

// One column of the walk.  (A, B): the column's record - bit p = the decision (cell_codes, dp_core.h) of the cell on the walker's
// diagonal band; rows above the matrix read as zero.  All the ups of the column (a run of set bits of A & ~B from bit p downwards),
// then the one move that leaves it (left or diagonal).  put(i, op): op number i of the walk (only while `store`).
// Returns 0: on to the column before; 1: out of the band (redo from (r, c)); 2: out of the matrix (the walk is over).
template <class Put>
NECAT_HD int band_walk_col(BandWalk& w, const u32 A, const u32 B, const int mlen, const bool store, Put& put)
{
    const u32 up = A & ~B;
    int run = clz32(~up << (31 - w.p));
    run = run < w.p + 1 ? run : w.p + 1;
    if (run) {
        if (store) for (int i = 0; i < run; ++i) put(w.n + i, 1);
        if (!w.hit) {
            // `run` non-matches in a row: the match run is 0 after the first (and "seen" only for a run length of 0)
            if (mlen == 0) { w.hit = 1; w.acnt = w.n + 1; w.qcnt = w.nq + 1; w.tcnt = w.nt; w.mcnt = w.nmat; }
            w.m = 0; w.nq += run;
        }
        w.n += run; w.r -= run; w.p -= run;
        if (w.r < 0) return 2;
        if (w.p < 0) return 1;
    }
    const u32 a = (A >> w.p) & 1u, b = (B >> w.p) & 1u;                  // not (1, 0): the ups are behind us
    const int left = (int)(b & (a ^ 1u)), mt = (int)((a | b) ^ 1u);
    if (store) put(w.n, (int)(a | (b << 1)));
    ++w.n; w.nmat += mt;
    if (!w.hit) {
        w.nq += 1 - left; w.nt += 1;
        w.m = mt ? w.m + 1 : 0;
        if (w.m == mlen) { w.hit = 1; w.acnt = w.n; w.qcnt = w.nq; w.tcnt = w.nt; w.mcnt = w.nmat; }
    }
    w.r -= 1 - left; w.c -= 1; w.p += left;
    if ((w.r | w.c) < 0) return 2;
    return w.p > 31 ? 1 : 0;
}

// band_walk_col written for the walker WAVE: 64 lanes on 64 different walks, so a data-dependent branch is executed by the whole wave anyway and every taken
// branch is a bubble in the one chain the workgroup waits for.  The moves are selects; the two things only some lanes do - keeping ops, and the tail scan of
// a walk that has not yet seen its run of matches - sit behind wave-uniform tests (NECAT_ANY).  No status code per column: a lane is `alive` until it leaves
// the band or the matrix, and which of the two it was is read off (r, c, p) after the segment (band_walk_why); `here`: the lane has reached this column
// (x <= its entry column; always true for a lane that entered the segment at its last column).  BW: diagonals per record (bits of A and of B).
template <int MAXOPS, int BW = 32, class Store>
NECAT_HD void band_walk_col3(BandWalk& w, bool& alive, const bool here, const u32 A, const u32 B, const int mlen, const bool store, Store& st_op, int& ovf)
{
    const bool act = alive && here;
    const u32 lim = (u32)w.p + 1u;
    u32 run = (u32)clz32((~A | B) << ((31u - (u32)w.p) & 31u));           // the ups below bit p (p in [0, BW) while the lane is alive) ..
    run = run < lim ? run : lim;                                          // .. ending at the band's bit 0
    run = act ? run : 0u;
    if (NECAT_ANY(run > 0 && (store || !w.hit))) {
        if (run > 0) {
            if (store) for (u32 i = 0; i < run; ++i) { if (w.n + (int)i < MAXOPS) st_op(w.n + (int)i, 1); else ovf = 1; }
            if (!w.hit) {
                if (mlen == 0) { w.hit = 1; w.acnt = w.n + 1; w.qcnt = w.nq + 1; w.tcnt = w.nt; w.mcnt = w.nmat; }
                w.m = 0; w.nq += (int)run;
            }
        }
    }
    w.n += (int)run; w.r -= (int)run; w.p -= (int)run;
    const bool go = act && (w.r | w.p) >= 0;
    const u32 p2 = (u32)w.p & 31u;
    const u32 a = (A >> p2) & 1u, b = (B >> p2) & 1u;
    const int left = (int)(b & (a ^ 1u)), nm = (int)(a | b);
    if (NECAT_ANY(go && store)) { if (go && store) { if (w.n < MAXOPS) st_op(w.n, (int)(a | (b << 1))); else ovf = 1; } }
    if (NECAT_ANY(go && !w.hit)) {
        if (go && !w.hit) {
            w.nq += 1 - left; w.nt += 1;
            w.m = nm ? 0 : w.m + 1;
            if (w.m == mlen) { w.hit = 1; w.acnt = w.n + 1; w.qcnt = w.nq; w.tcnt = w.nt; w.mcnt = w.nmat + 1 - nm; }
        }
    }
    if (go) { w.n += 1; w.nmat += 1 - nm; w.r -= 1 - left; w.c -= 1; w.p += left; }
    alive = act ? (go && (w.r | w.c) >= 0 && (u32)w.p < (u32)BW) : alive;
}
// after a segment, for a lane that was walking when it began: 0 = still walking (on to the segment before), 1 = left the band (redo from (r, c)), 2 = left the matrix
NECAT_HD int band_walk_why(const BandWalk& w, const bool alive) { return alive ? 0 : ((w.r | w.c) < 0 ? 2 : 1); }

}  // namespace necat
