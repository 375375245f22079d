// asm_kernels.h - the block aligner of oc2asmpm on the device: blockwise_edlib_align (asm_pm/blockwise_edlib.c:1205-1371 = onc_align with
// 2048-bp blocks, tail match length 8) for many (read, subject strand, anchor) triples at once.  First design, correctness first: ONE LANE runs
// one alignment from its anchor to both ends - the per-lane cores the CPU tests replay at this block size (ext_plan<2048>, myers_block<44 words>,
// traceback_block, ext_finish_block; tests/test_host_core.py::test_cores_at_block_2048_match_oracle) in the loop of check_core.cpp.  A wave's 64 lanes
// share a band slab ([column][word][lane] 16-byte records, coalesced when the lanes walk their columns together) and an op slab.  The column state
// (44 words x P, M, record, score + the query planes) does not fit the register file: it lives in scratch.  DESIGN 6h has what comes next (the
// cooperative, register-resident DP of the 512-bp stage at 32 / 44 words).
#pragma once
#include "dp_core.h"
#include "ext_core.h"

namespace necat {

constexpr int kAsmBlock = 2048;                                   // kOcaBlockSize2048 (blockwise_edlib.h:27, hbn_align.c:8)
constexpr int kAsmCols = (int)((kAsmBlock + 99) * 1.3);           // 2791: the longest last block (blockwise_edlib.c:1031-1036)
constexpr int kAsmWords = (kAsmCols + 63) / 64;                   // 44
constexpr int kAsmTW = (kAsmCols + 31) / 32 + 1;                  // target words of a block (+1: load32 of the last partial word)
constexpr int kAsmOps = 2 * kAsmCols + 8;                         // ops of one block alignment
constexpr size_t kAsmBandWave = (size_t)kAsmCols * kAsmWords * 64 * 16;     // bytes of one wave's band slab (126 MB)
constexpr size_t kAsmOpsWave = (size_t)kAsmOps * 64;

struct AsmAnchor { i32 q, s, sdir, qoff, soff; };        // local ids; qoff on the forward read, soff on strand sdir of the subject
struct AsmOut { i32 qoff, qend, toff, tend, cols, mat, lfrom, lto, rfrom, rto, blocks, err; };

struct AsmMat {
    ulonglong2* slab; int lane;
    NECAT_D bool skip_nw() const { return false; }
    NECAT_D void store(int c, int b, u64 A, u64 B) { slab[((size_t)c * kAsmWords + b) * 64 + lane] = make_ulonglong2(A, B); }
    NECAT_D void rec(int c, int b, u64& A, u64& B) const { const ulonglong2 v = slab[((size_t)c * kAsmWords + b) * 64 + lane]; A = v.x; B = v.y; }
};
struct AsmTgt { const u64* w; NECAT_D int code(int c) const { return (int)((w[c >> 5] >> ((c & 31) * 2)) & 3); } };
struct AsmOps {
    u8* ops; int overflow; TailScan ts;
    NECAT_D void push(int op) { if (ts.n < kAsmOps) ops[(size_t)ts.n * 64] = (u8)op; else overflow = 1; tail_push(ts, op); }
};
struct AsmROps { const u8* ops; NECAT_D int operator()(int j) const { return ops[(size_t)j * 64]; } };
struct AsmSame {
    const MyersRegs<kAsmWords>* R; const u64* tw;
    NECAT_D bool operator()(int i) const
    {
        const int q = (int)((~R->nlo[i >> 6] >> (i & 63)) & 1) | ((int)((~R->nhi[i >> 6] >> (i & 63)) & 1) << 1);
        return q == (int)((tw[i >> 5] >> ((i & 31) * 2)) & 3);
    }
};

#if NECAT_XCHECK          // the lane-per-alignment kernel the cooperative path replaced: cross-check build only (necat_hip.hip)
__global__ void __launch_bounds__(64)
k_asm_align(const AsmAnchor* __restrict__ anchors, u32 n, DevVolume reads, DevVolume ref, double error, int tail_match_len,
            char* __restrict__ band_pool, u8* __restrict__ ops_pool, u8* __restrict__ cols, const u64* __restrict__ cols_off, AsmOut* __restrict__ out)
{
    const u32 wave = blockIdx.x;
    const int lane = (int)threadIdx.x;
    const u32 i = wave * 64 + (u32)lane;
    if (i >= n) return;
    const AsmAnchor a = anchors[i];
    const i64 q_g0 = (i64)reads.seq_off[a.q], s_g0 = (i64)ref.seq_off[a.s];
    const i32 qlen = (i32)(reads.seq_off[a.q + 1] - reads.seq_off[a.q]), slen = (i32)(ref.seq_off[a.s + 1] - ref.seq_off[a.s]);
    ExtTask t;
    ext_init(t, (i32)i, 0, q_g0, qlen, s_g0, slen, a.qoff, a.soff);
    AsmMat mat; mat.slab = reinterpret_cast<ulonglong2*>(band_pool + (size_t)wave * kAsmBandWave); mat.lane = lane;
    u8* const ops = ops_pool + (size_t)wave * kAsmOpsWave + lane;
    u8* const my_cols = cols + cols_off[i];
    MyersRegs<kAsmWords> R;
    u64 tw[kAsmTW];
    int blocks = 0, err = 0;
    while (ext_plan<kAsmBlock>(t)) {
        if (++blocks > 4096) { err = 31; break; }
        // fragment geometry: element e of the query fragment is strand position QS + qidx + e (right) / QS - 1 - qidx - e (left) of the forward
        // read; the subject likewise on its strand, the reverse strand read backwards and complemented (packed_db.c:268-274)
        const bool right = t.phase == 1;
        const i64 qp = right ? (i64)t.QS + t.qidx : (i64)t.QS - 1 - t.qidx;
        const i64 q_base = q_g0 + qp; const int q_dir = right ? +1 : -1;
        const i64 sp = right ? (i64)t.TS + t.tidx : (i64)t.TS - 1 - t.tidx;
        const i64 t_base = a.sdir ? s_g0 + slen - 1 - sp : s_g0 + sp;
        const int t_dir = a.sdir ? (right ? -1 : +1) : (right ? +1 : -1), t_comp = a.sdir ? 1 : 0;
        const int qn = t.qblk, tn = t.tblk;
        if (qn > kAsmCols || tn > kAsmCols) { err = 32; break; }
        const int nblk = (qn + 63) >> 6;
        for (int b = 0; b < kAsmWords; ++b) {
            u64 lo = 0, hi = 0;
            if (b < nblk) load64_planes(reads.bases, q_base, q_dir, 0, b * 64, &lo, &hi);
            R.nlo[b] = ~lo; R.nhi[b] = ~hi;
        }
        for (int w = 0; w * 32 < tn; ++w) tw[w] = load32_dir(ref.bases, t_base + (i64)t_dir * (w * 32), t_dir, t_comp);
        AsmTgt tg; tg.w = tw;
        const MyersResult mr = myers_block<kAsmWords, false>(R, qn, tn, error, tg, mat);
        if (mr.err) { err = mr.err; break; }
        const int done = ext_block_done(t, mr.dist, mr.endc);
        AsmOps ow; ow.ops = ops; ow.overflow = 0;
        tail_init(ow.ts, done ? tail_match_len : kOcaMatCnt);
        if (mr.dist >= 0) traceback_block(qn, mr.endc + 1, mat, ow);
        if (ow.overflow) { err = 33; break; }
        AsmROps rd; rd.ops = ops;
        AsmSame same; same.R = &R; same.tw = tw;
        const int stream_at = t.phase == 1 ? t.s_lto : 0;
        const int nops = ow.ts.n;
        const ExtKept kept = ext_finish_block(t, mr.dist, mr.endc, done, ow.ts, rd, same);
        // the kept columns join the alignment stream of this extension: forward columns [0, cols) of the block's alignment (op r of the walk is
        // forward column nops - 1 - r), or `cols` matches (the exact-prefix fallback)
        u8* dst = my_cols + stream_at + kept.at;
        if (kept.exact) { for (int f = 0; f < kept.cols; ++f) dst[f] = 0; }
        else for (int f = 0; f < kept.cols; ++f) dst[f] = ops[(size_t)(nops - 1 - f) * 64];
    }
    AsmOut o;
    o.qoff = t.r_qoff; o.qend = t.r_qend; o.toff = t.r_toff; o.tend = t.r_tend; o.cols = t.r_cols; o.mat = t.r_mat;
    o.lfrom = t.s_lfrom; o.lto = t.s_lto; o.rfrom = t.s_rfrom; o.rto = t.s_rto; o.blocks = blocks; o.err = err;
    out[i] = o;
}
#endif

}  // namespace necat
