// ext_core.h - the block-stitching driver of onc_align (gapped_align/oc_aligner.c:303) restated as a
// per-candidate state machine that advances one 512-bp block per round:
//
//   plan  (ext_plan)         = get_next_sequence_block (oc_aligner.c:111-155) + the glue between the
//                              left and the right extension (oc_aligner.c:340-417)
//   DP + traceback           = Edlib_align (dp_core.h)
//   finish (ext_finish_block)= the tail trimming of oca_extend (oc_aligner.c:216-262)
//
// The reference materialises gapped strings and re-scans them; a candidate here only carries running
// column / base / match counters of its alignment stream (what M4 needs: coordinates + identity), so
// no strings ever leave the GPU.
#pragma once
#include "dev_common.h"

namespace necat {

struct ExtTask {
    // constants
    i32 cand;            // candidate index
    i32 qdir;
    i64 q_g0, s_g0;      // global base offsets of the query read (reads volume) / subject read (ref volume)
    i32 qlen, slen;
    // anchor + progress
    i32 QS, TS;          // current anchor (moved by the left extension, oc_aligner.c:356-357, :398-399)
    i32 phase;           // 0 = left extension, 1 = right extension, 2 = done
    i32 ext_done;        // current extension finished (a block set `done`)
    i32 ext_q, ext_t;    // sizes of the current extension (query_size/target_size of oca_extend)
    i32 qidx, tidx;
    // current block
    i32 qblk, tblk, last;
    // alignment-stream statistics of the current extension, in stream order
    i32 m_run, found;                        // running match count / first run of 8 seen
    i32 pre_cols, pre_q, pre_t, pre_mat;     // totals at the column closing that first run
    i32 tot_cols, tot_q, tot_t, tot_mat;
    // kept part of the left extension
    i32 l_cols, l_q, l_t, l_mat;
    // final result (oc_aligner.c:419-450)
    i32 r_qoff, r_qend, r_toff, r_tend, r_cols, r_mat;
    // alignment columns (only when the caller wants them, necat_onc_align_batch): both streams are written
    // to one per-task region in stream order, left stream at [0, s_lto), right stream from s_lto on; the
    // final alignment is region[s_lfrom, s_lto) reversed followed by region[s_lto + s_rfrom, s_lto + s_rto)
    i32 s_lfrom, s_lto, s_rfrom, s_rto;
    i32 sdir;            // strand of the SUBJECT (0 everywhere but in the overlapper of corrected reads, asm_pm: the query stays forward there)
    u64 ops_base;
};

NECAT_HD void ext_reset_stream(ExtTask& t)
{
    t.m_run = 0; t.found = 0;
    t.pre_cols = t.pre_q = t.pre_t = t.pre_mat = 0;
    t.tot_cols = t.tot_q = t.tot_t = t.tot_mat = 0;
    t.qidx = t.tidx = 0; t.ext_done = 0;
}

// calc_reference_range (reference_mapping/rm_worker.c:43-59): a read is mapped against the stretch of a reference sequence it can
// reach from the anchor - 1.3 x what is left of the read on either side, cut at the sequence ends.  from / to: the stretch,
// woff: the anchor inside it.
NECAT_HD void rm_window(i64 qoff, i64 qsize, i64 soff, i64 ssize, i64* from, i64* to, i64* woff)
{
    i64 n = qoff < soff ? (i64)((double)qoff * 1.3) : soff;
    if (n > soff) n = soff;
    *from = soff - n; *woff = n;
    const i64 sr = ssize - soff, qr = qsize - qoff;
    n = qr < sr ? (i64)((double)qr * 1.3) : sr;
    if (n > sr) n = sr;
    *to = soff + n;
}

NECAT_HD void ext_init(ExtTask& t, i32 cand, i32 qdir, i64 q_g0, i32 qlen, i64 s_g0, i32 slen, i32 qoff, i32 soff)
{
    t.cand = cand; t.qdir = qdir; t.q_g0 = q_g0; t.s_g0 = s_g0; t.qlen = qlen; t.slen = slen;
    t.QS = qoff; t.TS = soff; t.phase = 0;
    t.ext_q = qoff; t.ext_t = soff;           // left: oca_extend(query + QS - 1, QS, target + TS - 1, TS, ...)
    ext_reset_stream(t);
    t.qblk = t.tblk = t.last = 0;
    t.l_cols = t.l_q = t.l_t = t.l_mat = 0;
    t.r_qoff = t.r_qend = t.r_toff = t.r_tend = t.r_cols = t.r_mat = 0;
    t.s_lfrom = t.s_lto = t.s_rfrom = t.s_rto = 0;
    t.sdir = 0; t.ops_base = 0;
}

// one alignment column in stream order
NECAT_HD void ext_stream_col(ExtTask& t, bool is_match, bool has_q, bool has_t)
{
    t.tot_cols += 1; t.tot_q += has_q; t.tot_t += has_t; t.tot_mat += is_match;
    if (!t.found) {
        t.m_run = is_match ? t.m_run + 1 : 0;
        if (t.m_run == kOcaMatCnt) {
            t.found = 1;
            t.pre_cols = t.tot_cols; t.pre_q = t.tot_q; t.pre_t = t.tot_t; t.pre_mat = t.tot_mat;
        }
    }
}

// Decide the next block of the task, or finish it.  Returns true when a block (t.qblk x t.tblk) is
// scheduled; false when the task is done (results in t.r_*).
// BLOCK: the desired block size - kOcaBlockSize for onc_align, 2048 for its clone in asm_pm/blockwise_edlib.c
template <int BLOCK = kOcaBlockSize>
NECAT_HD bool ext_plan(ExtTask& t)
{
    for (;;) {
        if (t.phase == 2) return false;
        if (!t.ext_done) {
            // get_next_sequence_block (oc_aligner.c:111-155)
            const int qleft = t.ext_q - t.qidx, tleft = t.ext_t - t.tidx;
            int qblk, tblk, last;
            if (qleft < BLOCK + 100 || tleft < BLOCK + 100) {
                qblk = (int)((double)tleft * 1.3); if (qleft < qblk) qblk = qleft;
                tblk = (int)((double)qleft * 1.3); if (tleft < tblk) tblk = tleft;
                last = 1;
            } else { qblk = BLOCK; tblk = BLOCK; last = 0; }
            if (qblk != 0 && tblk != 0) { t.qblk = qblk; t.tblk = tblk; t.last = last; return true; }
        }
        // the current extension is over
        if (t.phase == 0) {
            // oc_aligner.c:340-367: keep the left alignment only beyond its first run of 8 matches
            t.s_lto = t.tot_cols; t.s_lfrom = t.tot_cols;
            if (t.found) {
                t.QS -= t.pre_q; t.TS -= t.pre_t;
                t.l_cols = t.tot_cols - t.pre_cols; t.l_q = t.tot_q - t.pre_q;
                t.l_t = t.tot_t - t.pre_t; t.l_mat = t.tot_mat - t.pre_mat;
                t.s_lfrom = t.pre_cols;
            }
            t.phase = 1;
            t.ext_q = t.qlen - t.QS; t.ext_t = t.slen - t.TS;   // oca_extend(query + QS, query_size - QS, ...)
            ext_reset_stream(t);
        } else {
            int f_cols = 0, f_q = 0, f_t = 0, f_mat = 0;
            t.s_rfrom = 0; t.s_rto = t.tot_cols;
            if (t.l_cols == 0) {
                t.s_rfrom = t.found ? t.pre_cols - kOcaMatCnt : t.tot_cols;
                // oc_aligner.c:386-416: no left part -> start the right alignment at its first run of 8
                if (t.found) {
                    t.QS += t.pre_q - kOcaMatCnt; t.TS += t.pre_t - kOcaMatCnt;
                    f_cols = t.tot_cols - (t.pre_cols - kOcaMatCnt);
                    f_q = t.tot_q - (t.pre_q - kOcaMatCnt);
                    f_t = t.tot_t - (t.pre_t - kOcaMatCnt);
                    f_mat = t.tot_mat - (t.pre_mat - kOcaMatCnt);
                }
            } else { f_cols = t.tot_cols; f_q = t.tot_q; f_t = t.tot_t; f_mat = t.tot_mat; }
            t.r_qoff = t.QS - t.l_q; t.r_qend = t.QS + f_q;
            t.r_toff = t.TS - t.l_t; t.r_tend = t.TS + f_t;
            t.r_cols = t.l_cols + f_cols; t.r_mat = t.l_mat + f_mat;
            t.phase = 2;
            return false;
        }
    }
}

// Fragment geometry of the scheduled block: element i of the query fragment is base
// (q_base + q_dir * i) of the reads volume (complemented when q_comp), same for the target.
struct FragGeom { i64 q_base; int q_dir, q_comp; i64 t_base; int t_dir, t_comp; };

NECAT_HD FragGeom ext_frag_geom(const ExtTask& t)
{
    FragGeom g;
    // strand position p of the query: left  p = QS - 1 - qidx - i ; right p = QS + qidx + i
    // FWD read: g = q_g0 + p ; REV read: g = q_g0 + qlen - 1 - p with complement (packed_db.c:268-274)
    const bool right = t.phase == 1;
    const i64 p0 = right ? (i64)t.QS + t.qidx : (i64)t.QS - 1 - t.qidx;
    if (t.qdir == 0) { g.q_base = t.q_g0 + p0; g.q_dir = right ? +1 : -1; g.q_comp = 0; }
    else { g.q_base = t.q_g0 + t.qlen - 1 - p0; g.q_dir = right ? -1 : +1; g.q_comp = 1; }
    const i64 s0 = right ? (i64)t.TS + t.tidx : (i64)t.TS - 1 - t.tidx;
    if (!t.sdir) { g.t_base = t.s_g0 + s0; g.t_dir = right ? +1 : -1; g.t_comp = 0; }
    else { g.t_base = t.s_g0 + t.slen - 1 - s0; g.t_dir = right ? -1 : +1; g.t_comp = 1; }      // reverse strand: read backwards, complemented
    return g;
}

// Running statistics of one block's alignment, gathered while the traceback emits ops END -> START:
// totals of the whole alignment, and the totals of its tail up to (and including) the first run of
// M consecutive matches met from the end (oc_aligner.c:224-241).  With these the kept part of the
// block is known in closed form and the op list never has to be read back.
struct TailScan {
    int M;                       // run length looked for (8, or tail_match_len on a final block)
    int n, nq, nt, nmat;         // whole alignment: columns, query bases, target bases, matches
    int m;                       // current run of matches (frozen once hit)
    int hit;                     // the run was found
    int acnt, qcnt, tcnt, mcnt;  // columns / bases / matches scanned when the run completed
};

NECAT_HD void tail_init(TailScan& s, int M) { s.M = M; s.n = s.nq = s.nt = s.nmat = 0; s.m = 0; s.hit = 0; s.acnt = s.qcnt = s.tcnt = s.mcnt = 0; }

NECAT_HD void tail_push(TailScan& s, int op)
{
    // branch-free: executed once per traceback step by every lane of the wave
    const int hq = op != 2, ht = op != 1, mt = op == 0;
    s.n += 1; s.nq += hq; s.nt += ht; s.nmat += mt;
    const int m2 = mt ? s.m + 1 : 0;
    const bool now = !s.hit && m2 == s.M;
    s.m = s.hit ? s.m : m2;
    s.acnt = now ? s.n : s.acnt; s.qcnt = now ? s.nq : s.qcnt; s.tcnt = now ? s.nt : s.tcnt; s.mcnt = now ? s.nmat : s.mcnt;
    s.hit |= now ? 1 : 0;
}

// Whether this block ends its extension, known before the traceback: Edlib_align aligns the whole
// query fragment (qfae = qn) and the target up to its end column (tfae = endc + 1), so the
// "> 30 unaligned bases on both sides" test of oc_aligner.c:221 can only fire on a failed block.
NECAT_HD int ext_block_done(const ExtTask& t, int dist, int endc)
{
    const int qfae = dist >= 0 ? t.qblk : 0, tfae = dist >= 0 ? endc + 1 : 0;
    int done = t.last;
    if (t.qblk - qfae > 30 && t.tblk - tfae > 30) done = 1;
    return done;
}

// Tail trimming + stream update after the block's alignment (oc_aligner.c:216-262).
//   dist < 0  : Edlib_align failed (empty alignment)
//   ts        : statistics gathered during the traceback with M = ext_block_done ? tail_match_len : 8
//   rops(j)   : op j of the alignment in END -> START order (only read while the stream has not yet
//               seen its first run of 8 matches, i.e. on the first block(s) of an extension)
//   same(i)   : query fragment element i == target fragment element i (exact-prefix fallback)
// What the block contributes to the stream, for callers that keep the alignment columns: `cols` columns
// starting at stream column `at`; they are forward columns [0, cols) of the block's alignment, or - exact
// (the exact-match fallback) - `cols` match columns.
struct ExtKept { int at, cols, exact; };

template <class ROps, class Same>
NECAT_HD ExtKept ext_finish_block(ExtTask& t, int dist, int endc, int done, const TailScan& ts, ROps& rops, Same& same)
{
    ExtKept kept; kept.at = t.tot_cols; kept.cols = 0; kept.exact = 0;
    const int qn = t.qblk, tn = t.tblk;
    const int n = dist >= 0 ? ts.n : 0;
    const int kfirst = n - ts.acnt;       // forward index where the tail scan stopped (k in the reference)
    if (dist < 0 || !ts.hit || kfirst < 1) {
        // exact-match prefix of the raw fragments, then stop (oc_aligner.c:243-254)
        const int lim = qn < tn ? qn : tn;
        for (int i = 0; i < lim; ++i) {
            if (!same(i)) break;
            ext_stream_col(t, true, true, true);
            ++kept.cols;
        }
        kept.exact = 1;
        done = 1;
    } else {
        const int M = ts.M;
        t.qidx += ts.nq - ts.qcnt; t.tidx += ts.nt - ts.tcnt;      // qfae - qcnt, tfae - tcnt
        // kept columns: forward [0, kfirst) plus, on a final block, the M matching columns (:258)
        int keep_cols = n - ts.acnt, keep_q = ts.nq - ts.qcnt, keep_t = ts.nt - ts.tcnt, keep_mat = ts.nmat - ts.mcnt;
        if (done) { keep_cols += M; keep_q += M; keep_t += M; keep_mat += M; }
        const int lo = done ? ts.acnt - M : ts.acnt;
        int jj = n - 1;
        // until the stream has its first run of 8 matches the columns must be replayed in order
        const int c0 = t.tot_cols, q0 = t.tot_q, t0 = t.tot_t, m0 = t.tot_mat;
        while (!t.found && jj >= lo) {
            const int op = rops(jj);
            ext_stream_col(t, op == 0, op != 2, op != 1);
            --jj;
        }
        // the rest only adds to the totals
        t.tot_cols = c0 + keep_cols; t.tot_q = q0 + keep_q; t.tot_t = t0 + keep_t; t.tot_mat = m0 + keep_mat;
        kept.cols = keep_cols;
    }
    t.ext_done = done;
    return kept;
}

}  // namespace necat
