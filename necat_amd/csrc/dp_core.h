// dp_core.h - block-wise banded Myers bit-vector alignment (the Edlib_align equivalent,
// gapped_align/edlib_ex.c:733): per-lane cores of the DP kernel (SHW distance pass + banded NW
// pass that stores the P/M/score band for the traceback) and of the traceback kernel.
//
// One GPU lane owns one block alignment.  The column state (P, M, score of each 64-row word) lives
// in registers: the word loop is fully unrolled over NW so every register index is static, the
// per-lane band [fblk, lblk] becomes a predicate, and words no lane of the wave needs are skipped
// with one wave-uniform ballot.  Query symbols are kept as two complemented bit-planes per word
// (Eq is rebuilt with 3 logic ops instead of a 4 x NW peq table).  Band growth / shrink rules,
// the k updates and the end-column choice follow the reference line by line, so distances, end
// columns and tracebacks are bit-identical.
#pragma once
#include "dev_common.h"
#include "ext_core.h"      // TailScan (walk_block)

namespace necat {

#if defined(__HIP_DEVICE_COMPILE__)
#define NECAT_ANY(cond) (__ballot(cond) != 0ULL)
#else
#define NECAT_ANY(cond) (cond)
#endif

constexpr u64 kHighBit = 1ULL << 63;

// edlib_ex.c:71-106 calculateBlock (Myers' Advance_Block), written on 32-bit halves: gfx950 has no
// full-rate 64-bit integer add / shift (v_lshl_add_u64, v_lshlrev_b64 issue at a fraction of the
// 32-bit rate and were ~1/3 of this function's time); add-with-carry and funnel shifts are full rate.
NECAT_HD int advance_block_full(u64 Pv, u64 Mv, u64 Eq, int hin, u64& PvOut, u64& MvOut, u64& PhOut, u64& D0Out)
{
    const u32 pl = (u32)Pv, ph = (u32)(Pv >> 32), ml = (u32)Mv, mh = (u32)(Mv >> 32);
    u32 el = (u32)Eq;
    const u32 eh = (u32)(Eq >> 32);
    const u32 neg = (u32)hin >> 31;                 // 1 iff hin == -1  (hinIsNeg)
    const u32 pos = (u32)(hin + 1) >> 1;            // 1 iff hin == +1
    const u32 xvl = el | ml, xvh = eh | mh;         // Xv = Eq | Mv  (before the hin fix-up of Eq)
    el |= neg;
    const u32 al = el & pl, ah = eh & ph;
    const u32 sl = al + pl;                         // ((Eq & Pv) + Pv)
    const u32 sh = ah + ph + (u32)(sl < al);
    const u32 xhl = (sl ^ pl) | el, xhh = (sh ^ ph) | eh;
    u32 Phl = ml | ~(xhl | pl), Phh = mh | ~(xhh | ph);
    u32 Mhl = pl & xhl, Mhh = ph & xhh;
    PhOut = ((u64)Phh << 32) | Phl;                 // bit r: D[r][c] - D[r][c-1] == +1
    D0Out = ((u64)(xhh | mh) << 32) | (xhl | ml);   // bit r: D[r][c] == D[r-1][c-1]  (Hyyro's D0 = Xh | Mv)
    const int hout = (int)(Phh >> 31) - (int)(Mhh >> 31);
    Phh = (Phh << 1) | (Phl >> 31); Phl = (Phl << 1) | pos;
    Mhh = (Mhh << 1) | (Mhl >> 31); Mhl = (Mhl << 1) | neg;
    const u32 ol = Mhl | ~(xvl | Phl), oh = Mhh | ~(xvh | Phh);
    const u32 nl = Phl & xvl, nh = Phh & xvh;
    PvOut = ((u64)oh << 32) | ol;
    MvOut = ((u64)nh << 32) | nl;
    return hout;
}

NECAT_HD int advance_block(u64 Pv, u64 Mv, u64 Eq, int hin, u64& PvOut, u64& MvOut)
{
    u64 ph, d0;
    return advance_block_full(Pv, Mv, Eq, hin, PvOut, MvOut, ph, d0);
}

// The traceback's decision at a cell, two bits per cell (one band record = these two words of a 64-row word):
//     (A, B) = (1, 0) up     : D[r][c] = D[r-1][c] + 1            (Pv; wins over left, edlib_ex.c:458)
//              (0, 1) left   : D[r][c] = D[r][c-1] + 1, not up    (Ph)
//              (0, 0) diagonal, match    : D[r][c] = D[r-1][c-1]  (D0)
//              (1, 1) diagonal, mismatch : D[r][c] = D[r-1][c-1] + 1
// Pv = the column's updated vertical deltas, Ph / D0 from the same update.
NECAT_HD void cell_codes(u64 Pv, u64 Ph, u64 D0, u64& A, u64& B)
{
    A = Pv | ~(Ph | D0);
    B = ~Pv & (Ph | ~D0);
}

// One column update that also yields the band record of the word.  With Pv/Mv the word's deltas BEFORE the
// update, Pv' after it and Xh as in advance_block_full (P and M are disjoint), cell_codes collapses to
//     A = Pv' | (Pv & ~Xh)        B = ~Pv' & (Mv | ~Xh)
// (Ph | D0 = Mv | Xh | ~Pv and Ph | ~D0 = Mv | ~Xh): one 3-input bit-op per half each.
NECAT_HD int advance_block_rec(u64 Pv, u64 Mv, u64 Eq, int hin, u64& PvOut, u64& MvOut, u64& A, u64& B)
{
    const u32 pl = (u32)Pv, ph = (u32)(Pv >> 32), ml = (u32)Mv, mh = (u32)(Mv >> 32);
    u32 el = (u32)Eq;
    const u32 eh = (u32)(Eq >> 32);
    const u32 neg = (u32)hin >> 31;
    const u32 pos = (u32)(hin + 1) >> 1;
    const u32 xvl = el | ml, xvh = eh | mh;
    el |= neg;
    const u32 al = el & pl, ah = eh & ph;
    const u32 sl = al + pl;
    const u32 sh = ah + ph + (u32)(sl < al);
    const u32 xhl = (sl ^ pl) | el, xhh = (sh ^ ph) | eh;
    u32 Phl = ml | ~(xhl | pl), Phh = mh | ~(xhh | ph);
    u32 Mhl = pl & xhl, Mhh = ph & xhh;
    const int hout = (int)(Phh >> 31) - (int)(Mhh >> 31);
    Phh = (Phh << 1) | (Phl >> 31); Phl = (Phl << 1) | pos;
    Mhh = (Mhh << 1) | (Mhl >> 31); Mhl = (Mhl << 1) | neg;
    const u32 ol = Mhl | ~(xvl | Phl), oh = Mhh | ~(xvh | Phh);
    const u32 nl = Phl & xvl, nh = Phh & xvh;
    PvOut = ((u64)oh << 32) | ol;
    MvOut = ((u64)nh << 32) | nl;
    A = ((u64)(oh | (ph & ~xhh)) << 32) | (ol | (pl & ~xhl));
    B = ((u64)(~oh & (mh | ~xhh)) << 32) | (~ol & (ml | ~xhl));
    return hout;
}

template <int NW>
struct MyersRegs {
    u64 P[NW], M[NW];
    u64 A[NW], B[NW];       // NW pass: the band record (cell_codes) of the current column's words
    int S[NW];
    u64 nlo[NW], nhi[NW];   // complemented query bit-planes (bit r = row 64*b + r)
};

struct MyersResult {
    int dist;       // edit distance, -1 = no alignment within k
    int endc;       // end_locations[0] (edlib_ex.c:770)
    int err;        // non-zero: internal disagreement between the two passes
    u32 words;      // word updates performed (SHW + NW), for the roofline report
};

// Functor contracts:
//   Tgt::code(c)                      -> 2-bit target code of column c
//   Mat::store(c, b, A, B)            -> band word b of column c (NW pass): the traceback's decision at each of its
//                                        64 cells (cell_codes) - all the traceback ever asks of a cell
template <int NW, bool FULL, class Tgt, class Mat>
NECAT_HD MyersResult myers_block(MyersRegs<NW>& R, int qn, int tn, double error, Tgt& tgt, Mat& mat)
{
    MyersResult res; res.dist = -1; res.endc = -1; res.err = 0; res.words = 0;
    const int nblk = FULL ? NW : (qn + 63) / 64;
    const int W = FULL ? 0 : nblk * 64 - qn;
    const u64 padmask = (FULL || W == 0) ? 0ULL : (~0ULL << ((64 - W) & 63));    // build_peq pad bits (edlib_ex.c:46)
    int k = (int)((double)(qn < tn ? qn : tn) * error * 1.1);             // edlib_ex.c:751

#define NECAT_EQ(b, ma, mb) (((R.nlo[b] ^ (ma)) & (R.nhi[b] ^ (mb))) | ((!FULL && (b) == nblk - 1) ? padmask : 0ULL))

    // ------------------------------------------------------------------ SHW pass (edlib_ex.c:108-223)
    // Result of the pass: the smallest bottom-row value D[qn-1][c] <= k over all columns and the FIRST
    // column attaining it.  Ukkonen's band only prunes cells that cannot be <= k, so computing every
    // word of every column yields the same (distance, end column) - and costs less here: 64 lanes with
    // 64 different bands spent more on band bookkeeping and exec-mask traffic than the skipped words
    // saved (measured: 323 k VALU + 136 k SALU per wave banded).  The NW pass below keeps the
    // reference's banding because its band is what gets stored.
    int fblk = 0, lblk = 0;
#pragma unroll
    for (int b = 0; b < NW; ++b) { R.P[b] = ~0ULL; R.M[b] = 0ULL; }
    int best = -1, end0 = -1;
    int Slast = nblk * 64;
    for (int c = 0; c < tn; ++c) {
        const int tc = tgt.code(c);
        const u64 ma = (tc & 1) ? ~0ULL : 0ULL, mb = (tc & 2) ? ~0ULL : 0ULL;
        int hout = 1;
        res.words += (u32)nblk;
#pragma unroll
        for (int b = 0; b < NW; ++b) {
            if (FULL || NECAT_ANY(b < nblk)) {
                const u64 eq = NECAT_EQ(b, ma, mb);
                if (FULL || b < nblk) hout = advance_block(R.P[b], R.M[b], eq, hout, R.P[b], R.M[b]);
            }
        }
        Slast += hout;                      // hout of the last word = horizontal delta of the bottom row
        if (Slast <= k && (best == -1 || Slast <= best)) {
            if (Slast != best) { best = Slast; k = best; end0 = c - W; }
        }
    }
    if (!FULL && W > 0) {
        // edlib_ex.c:205-219: the last W true columns sit inside the last word
        u64 P = 0, M = 0;
#pragma unroll
        for (int b = 0; b < NW; ++b) if (b == nblk - 1) { P = R.P[b]; M = R.M[b]; }
        int score = Slast;
        for (int i = 0; i < W; ++i) {
            // scores[i + 1]: after consuming bit (63 - i)
            if (P & (kHighBit >> i)) --score;
            if (M & (kHighBit >> i)) ++score;
            if (score <= k && (best == -1 || score <= best)) {
                if (score != best) { k = best = score; end0 = tn - W + i; }
            }
        }
    }
    if (best == -1) return res;
    if (mat.skip_nw()) { res.dist = best; res.endc = end0; return res; }     // profiling-only switch

    // ------------------------------------------------------------------ NW pass (edlib_ex.c:226-370)
    const int d = best;
    const int tn2 = end0 + 1;
    k = d;
    { int ad = tn2 - qn; if (ad < 0) ad = -ad; if (k < ad) { res.err = 1; return res; } }
    { int mx = qn > tn2 ? qn : tn2; if (k > mx) k = mx; }
    fblk = 0;
    { int X = (k + qn - tn2) / 2; int Y = k < X ? k : X; int Z = (Y + 1 + 63) / 64; lblk = (nblk < Z ? nblk : Z) - 1; }
#pragma unroll
    for (int b = 0; b < NW; ++b) { R.S[b] = (b + 1) * 64; R.P[b] = ~0ULL; R.M[b] = 0ULL; }
    bool alive = true;
    for (int c = 0; c < tn2; ++c) {
        const int tc = tgt.code(c);
        const u64 ma = (tc & 1) ? ~0ULL : 0ULL, mb = (tc & 2) ? ~0ULL : 0ULL;
        int hout = 1, lastS = 0, firstS = 0;
        const int lblk0 = lblk;
        bool kdone = false;
        res.words += (u32)(lblk - fblk + 1);
#pragma unroll
        for (int b = 0; b < NW; ++b) {
            const bool act = (b >= fblk) & (b <= lblk0);
            const bool edge = (b == lblk0 + 1) & (b < nblk);
            if (NECAT_ANY(act | edge)) {
                const u64 eq = NECAT_EQ(b, ma, mb);
                if (act) {
                    hout = advance_block_rec(R.P[b], R.M[b], eq, hout, R.P[b], R.M[b], R.A[b], R.B[b]);
                    R.S[b] += hout; lastS = R.S[b];
                    if (b == fblk) firstS = R.S[b];
                } else if (edge) {
                    // k update of edlib_ex.c:297-302 (lblk0 < nblk - 1 here, so no W term)
                    { int X1 = tn2 - c - 1, X2 = qn - ((1 + lblk0) * 64 - 1) - 1; int Z = (X1 > X2 ? X1 : X2) + lastS; if (Z < k) k = Z; }
                    kdone = true;
                    const bool r = (lblk0 + 1) * 64 - 1 > k - lastS + 2 * 64 - 2 - tn2 + c + qn;   // edlib_ex.c:305
                    if (!r) {
                        u64 p, m;
                        const int nh = advance_block_rec(~0ULL, 0ULL, eq, hout, p, m, R.A[b], R.B[b]);
                        R.P[b] = p; R.M[b] = m;
                        R.S[b] = lastS - hout + 64 + nh;
                        lastS = R.S[b]; lblk = b; hout = nh; ++res.words;
                    }
                }
            }
        }
        if (!kdone) {
            int X1 = tn2 - c - 1, X2 = qn - ((1 + lblk0) * 64 - 1) - 1;
            int Z = (X1 > X2 ? X1 : X2) + ((lblk0 == nblk - 1) ? W : 0) + lastS;
            if (Z < k) k = Z;
        }
        {
            const bool need = lastS >= k + 64 || ((lblk + 1) * 64 - 1 > k - lastS + 2 * 64 - 2 - tn2 + c + qn + 1);
            if (NECAT_ANY(need)) {
#pragma unroll
                for (int b = NW - 1; b >= 0; --b)
                    if (b == lblk && lblk >= fblk &&
                        (R.S[b] >= k + 64 || ((b + 1) * 64 - 1 > k - R.S[b] + 2 * 64 - 2 - tn2 + c + qn + 1))) --lblk;
            }
        }
        {
            const bool need = firstS >= k + 64 || ((fblk + 1) * 64 - 1 < firstS - k - tn2 + qn + c);
            if (NECAT_ANY(need)) {
#pragma unroll
                for (int b = 0; b < NW; ++b)
                    if (b == fblk && fblk <= lblk &&
                        (R.S[b] >= k + 64 || ((b + 1) * 64 - 1 < R.S[b] - k - tn2 + qn + c))) ++fblk;
            }
        }
        if (lblk < fblk) { alive = false; break; }
#pragma unroll
        for (int b = 0; b < NW; ++b) {
            const bool in = (b >= fblk) & (b <= lblk);
            if (NECAT_ANY(in)) { if (in) mat.store(c, b, R.A[b], R.B[b]); }
        }
    }
    int d2 = -1;
    if (alive && lblk == nblk - 1) {
        u64 P = 0, M = 0; int S = 0;
#pragma unroll
        for (int b = 0; b < NW; ++b) if (b == nblk - 1) { P = R.P[b]; M = R.M[b]; S = R.S[b]; }
        int cs = S;
        if (W > 0) cs = S - popc64(P >> ((64 - W) & 63)) + popc64(M >> ((64 - W) & 63));   // calc_block_cell_scores()[W]
        if (cs <= k) d2 = cs;
    }
    if (d2 != d) { res.err = 2; return res; }     // edlib_ex.c:779-780 asserts the same
    res.dist = d; res.endc = end0;
#undef NECAT_EQ
    return res;
}

// ---------------------------------------------------------------------------------------------
// Traceback (edlib_ex.c:383-621, obtainAlignmentTraceback): move priority up > left > diagonal.
//
// The reference compares the scores of the three neighbours with the current score (carrying shifted
// copies of two band words and cached neighbour scores through the walk).  In an exact DP those
// comparisons are bit tests on the current cell alone:
//     up   <=> D[r][c] = D[r-1][c] + 1  <=> bit r of Pv(c)        (positive vertical delta)
//     left <=> D[r][c] = D[r][c-1] + 1  <=> bit r of Ph(c)        (positive horizontal delta)
//     else diagonal: the cell's value can only come from the diagonal then, a match <=> bit r of D0(c)
//     (diagonal delta 0) <=> the two bases are equal.
// The band only ever leaves out cells that cannot lie on an alignment of cost <= the block's distance, and
// every cell the walk stands on lies on one, so "is the neighbour inside the band" never decides anything
// (the reference's availability flags, edlib_ex.c:431-451) - checked on every block of the E. coli workload
// against the oracle's score-based walk (tests/test_host_core.py on the CPU, the GPU parity tests).
// The DP kernels fold the three vectors into the walk's decision, two bits per cell (cell_codes): a band
// record is 16 bytes, needs no validity tag, a step reads one record and never touches the sequences.
//   Mat::rec(c, b, A, B)     band word (c, b)
//   Ops::push(op)            receives ops in END -> START order
// Op codes: 0 match, 1 insert (consumes a query base), 2 delete (consumes a target base),
// 3 mismatch (edlib_ex.c:10-13).
// ---------------------------------------------------------------------------------------------
template <class Mat, class Ops>
NECAT_HD void traceback_block(int qn, int tn, Mat& mat, Ops& ops)
{
    // The step is written with selects, not nested branches: the lanes of a wave are on 64 different
    // paths, so every divergent region would be executed by the whole wave on every step anyway.
    const int nblk = (qn + 63) / 64, W = nblk * 64 - qn;
    int c = tn - 1, b = nblk - 1, pos = 63 - W;
    u64 A, B;
    mat.rec(c, b, A, B);
    int term = 0, term_op = 0;       // how the walk ended (boundary cases push runs of ops)
    for (;;) {
        const bool a1 = (A >> pos) & 1ULL, b1 = (B >> pos) & 1ULL;
        const bool go_up = a1 && !b1, go_left = !a1 && b1;               // up > left > diagonal, decided by the DP kernel
        const int op = go_up ? 1 : (go_left ? 2 : (a1 ? 3 : 0));
        const bool drow = !go_left, dcol = !go_up;                       // consumes a query base / a target base
        const bool cross = drow && pos == 0;                             // leaves the 64-row word upwards
        c -= dcol ? 1 : 0;
        int t = 0;
        if (cross && b == 0) t = go_up ? 1 : 4;                          // out of the first row
        if (dcol && c == -1) t = go_left ? 2 : 3;                        // out of the first column (tested first, edlib_ex.c)
        if (t) { term = t; term_op = op; break; }
        pos = drow ? (cross ? 63 : pos - 1) : pos;
        b -= cross ? 1 : 0;
        if (dcol || cross) mat.rec(c, b, A, B);
        ops.push(op);
    }
    if (term == 1) {                 // up move out of the first row
        ops.push(1);
        for (int i = 0; i < c + 1; ++i) ops.push(2);
    } else if (term == 2) {          // left move out of the first column
        ops.push(2);
        const int numUp = b * 64 + pos + 1;
        for (int i = 0; i < numUp; ++i) ops.push(1);
    } else if (term == 3) {          // diagonal move out of the first column
        ops.push(term_op);
        const int numUp = b * 64 + pos;
        for (int i = 0; i < numUp; ++i) ops.push(1);
    } else if (term == 4) {          // diagonal move out of the first row
        ops.push(term_op);
        for (int i = 0; i < c + 1; ++i) ops.push(2);
    }
}

// ---------------------------------------------------------------------------------------------
// The same walk, restated for the GPU's instruction mix (this is what k_traceback runs; traceback_block above is the
// reference formulation the CPU tests replay both against).  With r the query row and c the column of the cell, the two
// record bits of the cell ARE the op code (a | b << 1: 0 match, 1 up / insert, 2 left / delete, 3 mismatch); a move
// consumes a row unless it is "left" and a column unless it is "up"; the walk ends when a row or a column runs out and
// the rest of the other one is one run of inserts / deletes.  The tail statistics (TailScan) are only kept up step by step
// until the first run of M matches has been seen - a few dozen steps - after that a step only counts matches: every row
// and every column is consumed exactly once, so nq = qn and nt = tn at the end, and n is the step count.
//   Mat::rec(c, b, A, B)     band word (c, b)
//   Sink::put(i, op)         op number i (END -> START order), called only while Sink::storing()
// ---------------------------------------------------------------------------------------------

template <class Mat, class Sink>
NECAT_HD void walk_block(int qn, int tn, Mat& mat, Sink& sink, TailScan& ts)
{
    const int M = ts.M;
    const bool storing = sink.storing();
    int r = qn - 1, c = tn - 1, wb = r >> 6;
    u64 A, B;
    mat.rec(c, wb, A, B);
    int n = 0, nmat = 0;
    int m = 0, hit = 0, nq = 0, nt = 0, acnt = 0, qcnt = 0, tcnt = 0, mcnt = 0;
    for (;;) {
        const int sh = r & 63;
        const u32 a = (u32)(A >> sh) & 1u, b = (u32)(B >> sh) & 1u;
        const int op = (int)(a | (b << 1));
        const int drow = 1 - (int)(b & (a ^ 1u)), dcol = 1 - (int)(a & (b ^ 1u));    // not "left" / not "up"
        const int mt = (int)((a | b) ^ 1u);
        if (storing) sink.put(n, op);
        ++n; nmat += mt;
        if (!hit) {
            nq += drow; nt += dcol;
            m = mt ? m + 1 : 0;
            if (m == M) { hit = 1; acnt = n; qcnt = nq; tcnt = nt; mcnt = nmat; }
        }
        r -= drow; c -= dcol;
        if ((r | c) < 0) break;
        const int wb2 = r >> 6;
        if (dcol | (wb2 ^ wb)) mat.rec(c, wb2, A, B);
        wb = wb2;
    }
    // out of the first column: the rows left are inserts; out of the first row: the columns left are deletes
    const int kop = c < 0 ? 1 : 2, k = c < 0 ? r + 1 : c + 1;
    if (storing) for (int i = 0; i < k; ++i) sink.put(n + i, kop);
    n += k;
    if (!hit && k > 0) m = 0;
    ts.n = n; ts.nq = qn; ts.nt = tn; ts.nmat = nmat; ts.m = m; ts.hit = hit;
    ts.acnt = acnt; ts.qcnt = qcnt; ts.tcnt = tcnt; ts.mcnt = mcnt;
}

}  // namespace necat
