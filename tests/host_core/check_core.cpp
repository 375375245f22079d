// check_core.cpp - TEST ONLY.  Compiles the product's per-lane kernel cores (necat_amd/csrc/*_core.h)
// with g++ and replays them lane by lane on the CPU against the oracle, so kernel logic can be
// validated on a machine without a GPU.  Nothing here is part of the product path.
//
// usage: check_core <wrk_dir> <vid> k z q b s n a e [max_reads]
//
// -DCHECK_BLOCK=2048 -DCHECK_TAIL=8: the same replay with the block size and tail match length of the block aligner's clone in
// asm_pm/blockwise_edlib.c (oc2asmpm): the cores are generic in both, and the 2048-bp device path is built from them.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#include "../../necat_amd/csrc/seed_core.h"
#include "../../necat_amd/csrc/dp_core.h"
#include "../../necat_amd/csrc/ext_core.h"
extern "C" {
#include "../../oracle/necat_oracle.h"
}
using namespace necat;

#ifndef CHECK_BLOCK
#define CHECK_BLOCK 512
#endif
#ifndef CHECK_TAIL
#define CHECK_TAIL 1
#endif
constexpr int kBlk = CHECK_BLOCK, kNwFull = kBlk / 64, kNwMax = (int)((kBlk + 99) * 1.3) / 64 + 1, kTail = CHECK_TAIL;
static_assert(kBlk % 64 == 0, "block size");

struct HostVol { std::vector<u64> words; std::vector<u64> off; DevVolume dv; };

static void make_vol(const ora_volume& v, HostVol& h)
{
    const int G = 4;
    u64 nw = (v.nbases + 31) / 32;
    h.words.assign(nw + 2 * G, 0);
    for (u64 i = 0; i < v.nbases; ++i) {
        u64 c = (v.pac[i >> 2] >> ((~i & 3) << 1)) & 3;
        h.words[G + (i >> 5)] |= c << ((i & 31) * 2);
    }
    h.off.resize(v.nseq + 1);
    for (u64 i = 0; i < v.nseq; ++i) h.off[i] = v.offset[i];
    h.off[v.nseq] = v.nbases;
    h.dv.bases = h.words.data() + G; h.dv.seq_off = h.off.data(); h.dv.nbases = v.nbases; h.dv.nseq = v.nseq;
}

// ---- host functors for the DP core (lane stride 1)
struct HTgt { const u64* w; int code(int c) { return (int)((w[c >> 5] >> ((c & 31) * 2)) & 3); } };
struct HMat {
    std::vector<u64> Pv, Ph; int nw;
    void init(int cols, int nw_) { nw = nw_; Pv.assign((size_t)cols * nw, 0); Ph.assign((size_t)cols * nw, 0); }
    void store(int c, int b, u64 pv, u64 ph) { Pv[(size_t)c * nw + b] = pv; Ph[(size_t)c * nw + b] = ph; }
    bool skip_nw() const { return false; }
};
struct HMatR {
    const HMat* m;
    void rec(int c, int b, u64& Pv, u64& Ph) const { Pv = m->Pv[(size_t)c * m->nw + b]; Ph = m->Ph[(size_t)c * m->nw + b]; }
};
// -DCHECK_WINDOW: the band records the walk can reach.  The block's distance d and end column e are known after the SHW pass; a cell (r, c) of an
// alignment of cost d from (0, 0) to (qn - 1, e) has |r - c| + |(qn - 1 - e) - (r - c)| <= d, so the NW pass only needs to keep, per column, the
// words that meet rows c + ceil((D - d) / 2) .. c + floor((D + d) / 2), D = qn - 1 - e.  HMatW keeps exactly those; a read of anything else is counted.
struct HMatW {
    std::vector<u64> A, B; std::vector<char> have; int nw = 0, qn = 0, dlo = 0, dhi = 0;
    long stored = 0, offered = 0, bad_reads = 0;
    int cur_col = -1, cur_n = 0, max_per_col = 0;
    void init(int cols, int nw_, int qn_, int dist, int endc)
    {
        nw = nw_; qn = qn_;
        const int D = qn - 1 - endc;
        dlo = -((dist - D) / 2); dhi = (D + dist) / 2;
        A.assign((size_t)cols * nw, 0); B.assign((size_t)cols * nw, 0); have.assign((size_t)cols * nw, 0);
    }
    bool in_window(int c, int b) const
    {
        const int lo = std::max(0, c + dlo), hi = std::min(qn - 1, c + dhi);
        return lo <= hi && b >= (lo >> 6) && b <= (hi >> 6);
    }
    void store(int c, int b, u64 a, u64 bb) { ++offered; if (c != cur_col) { cur_col = c; cur_n = 0; } if (++cur_n > max_per_col) max_per_col = cur_n; if (!in_window(c, b)) return; ++stored; A[(size_t)c * nw + b] = a; B[(size_t)c * nw + b] = bb; have[(size_t)c * nw + b] = 1; }
    bool skip_nw() const { return false; }
    void rec(int c, int b, u64& a, u64& bb) { if (!have[(size_t)c * nw + b]) ++bad_reads; a = A[(size_t)c * nw + b]; bb = B[(size_t)c * nw + b]; }
};
// The compact slab the device path wants: [column][word - first stored word of the column], kSlots per column, one byte per column for the first
// word.  A block whose band is wider than kSlots somewhere is counted as an overflow (the device would redo it on a full slab).
struct HMatC {
    static constexpr int kSlots = 20;
    std::vector<u64> A, B; std::vector<int> first; int overflow = 0; long bad_reads = 0;
    void init(int cols) { A.assign((size_t)cols * kSlots, 0); B.assign((size_t)cols * kSlots, 0); first.assign((size_t)cols, -1); overflow = 0; }
    void store(int c, int b, u64 a, u64 bb)
    {
        if (first[(size_t)c] < 0) first[(size_t)c] = b;
        const int s = b - first[(size_t)c];
        if (s < 0 || s >= kSlots) { overflow = 1; return; }
        A[(size_t)c * kSlots + s] = a; B[(size_t)c * kSlots + s] = bb;
    }
    bool skip_nw() const { return false; }
    void rec(int c, int b, u64& a, u64& bb)
    {
        const int s = first[(size_t)c] < 0 ? -1 : b - first[(size_t)c];
        if (s < 0 || s >= kSlots) { ++bad_reads; a = bb = 0; return; }
        a = A[(size_t)c * kSlots + s]; bb = B[(size_t)c * kSlots + s];
    }
};
static long g_cmp_overflow = 0, g_cmp_bad = 0, g_cmp_diff = 0;
static long g_win_stored = 0, g_win_offered = 0, g_win_bad = 0, g_win_diff = 0, g_win_cols = 0, g_win_hist[64] = {0};
struct HOps { std::vector<int> v; TailScan ts; void push(int op) { v.push_back(op); tail_push(ts, op); } };
// the GPU formulation of the same walk (walk_block): must give the same ops and the same tail statistics
struct HSink { std::vector<int> v; bool storing() const { return true; } void put(int i, int op) { if ((int)v.size() <= i) v.resize(i + 1, -1); v[i] = op; } };
static long g_bad_walk = 0;
template <class M>
static void check_walk(int qn, int tn, M& m, const HOps& ref)
{
    HSink sk; TailScan ts; tail_init(ts, ref.ts.M);
    walk_block(qn, tn, m, sk, ts);
    const TailScan& a = ref.ts;
    const bool same = sk.v == ref.v && ts.n == a.n && ts.nq == a.nq && ts.nt == a.nt && ts.nmat == a.nmat && ts.hit == a.hit &&
                      ts.acnt == a.acnt && ts.qcnt == a.qcnt && ts.tcnt == a.tcnt && ts.mcnt == a.mcnt && (ts.hit || ts.m == a.m);
    if (!same) { if (g_bad_walk < 5) fprintf(stderr, "WALK MISMATCH %d x %d: n %d/%d nq %d/%d nt %d/%d nmat %d/%d hit %d/%d acnt %d/%d\n", qn, tn, ts.n, a.n, ts.nq, a.nq,
                 ts.nt, a.nt, ts.nmat, a.nmat, ts.hit, a.hit, ts.acnt, a.acnt); ++g_bad_walk; }
}

template <int NW, bool FULL>
static MyersResult run_block(const DevVolume& reads, const DevVolume& ref, const FragGeom& g, int qn, int tn, double error, HMat& mat, u64* tw, MyersRegs<NW>& R)
{
    for (int b = 0; b < NW; ++b) {
        u64 lo = 0, hi = 0;
        if (b * 64 < qn) load64_planes(reads.bases, g.q_base, g.q_dir, g.q_comp, b * 64, &lo, &hi);
        R.nlo[b] = ~lo; R.nhi[b] = ~hi;
    }
    for (int w = 0; w * 32 < tn; ++w) tw[w] = load32_dir(ref.bases, g.t_base + (i64)g.t_dir * (w * 32), g.t_dir, 0);
    HTgt tg; tg.w = tw;
    mat.init(tn, NW);
    return myers_block<NW, FULL>(R, qn, tn, error, tg, mat);
}

struct HRops { const std::vector<int>* v; int operator()(int j) const { return (*v)[j]; } };
template <int NW> struct HEq {
    const MyersRegs<NW>* R; const u64* tw;
    bool operator()(int row, int c) const {
        int q = (int)((~R->nlo[row >> 6] >> (row & 63)) & 1) | ((int)((~R->nhi[row >> 6] >> (row & 63)) & 1) << 1);
        return q == (int)((tw[c >> 5] >> ((c & 31) * 2)) & 3);
    }
};
template <int NW> struct HSame {
    const MyersRegs<NW>* R; const u64* tw;
    bool operator()(int i) const {
        int q = (int)((~R->nlo[i >> 6] >> (i & 63)) & 1) | ((int)((~R->nhi[i >> 6] >> (i & 63)) & 1) << 1);
        return q == (int)((tw[i >> 5] >> ((i & 31) * 2)) & 3);
    }
};

int main(int argc, char** argv)
{
    if (argc < 11) { fprintf(stderr, "usage: %s wrk_dir vid k z q b s n a e [max_reads] [job]\n", argv[0]); return 2; }
    const char* wrk = argv[1]; int vid = atoi(argv[2]);
    ora_options opt; ora_options_default(&opt);
    opt.kmer_size = atoi(argv[3]); opt.scan_window = atoi(argv[4]); opt.kmer_cnt_cutoff = atoi(argv[5]); opt.block_size = atoi(argv[6]);
    opt.block_score_cutoff = atoi(argv[7]); opt.num_candidates = atoi(argv[8]); opt.align_size_cutoff = atoi(argv[9]); opt.error = atof(argv[10]);
    int max_reads = argc > 11 ? atoi(argv[11]) : 1 << 30;
    opt.job = argc > 12 ? atoi(argv[12]) : 1;
    ora_volumes_info vi;
    if (ora_volumes_info_load(wrk, &vi)) return 2;
    ora_volume ref;
    if (ora_volume_load(vi.names[vid], &ref)) return 2;
    ora_index* ix = ora_index_build(&ref, opt.kmer_size, opt.kmer_cnt_cutoff);
    HostVol href; make_vol(ref, href);
    long bad_seed = 0, bad_ext = 0, n_cand = 0, n_blocks = 0, n_m4 = 0;
    for (int v = vid; v < vi.num_volumes; ++v) {
        ora_volume rd_own; const ora_volume* rd = &ref; HostVol hrd_own; const HostVol* hrd = &href;
        if (v != vid) { if (ora_volume_load(vi.names[v], &rd_own)) return 2; rd = &rd_own; make_vol(rd_own, hrd_own); hrd = &hrd_own; }
        ora_wfd* w = ora_wfd_new(ref.nbases, opt.block_size, opt.kmer_size, opt.block_score_cutoff);
        ora_aligner* al = ora_aligner_new(opt.error);
        ora_can_vec oc = {0, 0, 0};
        std::vector<uint8_t> fwd, rev, subj;
        SeedParams P; P.k = opt.kmer_size; P.z = opt.scan_window; P.block_size = opt.block_size; P.s_cutoff = opt.block_score_cutoff;
        P.align_cutoff = opt.align_size_cutoff; P.num_candidates = opt.num_candidates; P.job = opt.job; P.pairwise = 1;
        P.read_start_id = vi.read_start_id[v]; P.ref_start_id = vi.read_start_id[vid]; P.debug_phase = 0;
        const int H = 1 << 20;
        std::vector<u64> htab(4 * H, kHtEmpty); std::vector<SBlock> pool(H); std::vector<u64> cs(H + 1), uu(H + 1);
        std::vector<i32> f(H + 1), p(H + 1), t(H + 1), vv(H + 1); std::vector<DevCand> lcan(H + 1), outc(H);
        SeedScratch S; S.ht = htab.data(); S.ht_mask = 4 * H - 1; S.pool = pool.data(); S.pool_cap = H;
        S.cs = cs.data(); S.f = f.data(); S.p = p.data(); S.t = t.data(); S.v = vv.data(); S.u = uu.data(); S.lcan = lcan.data(); S.cs_cap = H + 1;
        S.out = outc.data(); S.out_cap = H;
        static MyersRegs<kNwFull> R8; static MyersRegs<kNwMax> R13; HMat mat; static u64 tw[2 * kNwMax + 2];
        int nreads = (int)std::min<u64>(rd->nseq, (u64)max_reads);
        for (int r = 0; r < nreads; ++r) {
            size_t L = rd->size[r];
            fwd.resize(L + 1); rev.resize(L + 1);
            oc.n = 0;
            ora_volume_extract(rd, r, 0, fwd.data());
            ora_find_candidates(fwd.data(), (int)L, r, 0, P.read_start_id, P.ref_start_id, 1, &ref, ix, &opt, w, &oc);
            ora_volume_extract(rd, r, 1, rev.data());
            ora_find_candidates(rev.data(), (int)L, r, 1, P.read_start_id, P.ref_start_id, 1, &ref, ix, &opt, w, &oc);
            // oracle per-read post-processing (pm_worker.c:133-140 / :163-171)
            std::vector<ora_candidate> ocv(oc.a, oc.a + oc.n);
            auto before = [](const ora_candidate& a, const ora_candidate& b) {
                if (a.score != b.score) return a.score > b.score; if (a.qdir != b.qdir) return a.qdir < b.qdir;
                if (a.sid != b.sid) return a.sid < b.sid; if (a.qoff != b.qoff) return a.qoff < b.qoff; return a.soff < b.soff; };
            if (opt.job == 1 || (int)ocv.size() > opt.num_candidates) { std::sort(ocv.begin(), ocv.end(), before); if ((int)ocv.size() > opt.num_candidates) ocv.resize(opt.num_candidates); }
            IndexView iv; iv.dense = ix->kmer_stats; iv.words = nullptr; iv.compact = nullptr;
            int n = seed_one_read(href.dv, iv, ix->offset_list, hrd->dv, r, P, S);
            bool same = n == (int)ocv.size();
            for (int i = 0; same && i < n; ++i) {
                const DevCand& a = outc[i]; const ora_candidate& b = ocv[i];
                same = a.qid == b.qid && a.sid == b.sid && a.qdir == b.qdir && a.score == b.score && a.qbeg == b.qbeg && a.qend == b.qend &&
                       a.qsize == b.qsize && a.sbeg == b.sbeg && a.send == b.send && a.ssize == b.ssize && a.qoff == b.qoff && a.soff == b.soff;
            }
            if (!same) { if (bad_seed < 5) fprintf(stderr, "SEED MISMATCH read %d: core %d vs oracle %zu\n", r, n, ocv.size()); ++bad_seed; }
            n_cand += n;
            if (opt.job != 1) continue;
            // extension of every candidate with both implementations
            for (size_t ci = 0; ci < ocv.size(); ++ci) {
                const ora_candidate& c = ocv[ci];
                subj.resize((size_t)c.ssize + 1);
                ora_volume_extract(&ref, (uint64_t)c.sid, 0, subj.data());
                ora_align_result ar;
                int ok = ora_onc_align(al, c.qdir == 0 ? fwd.data() : rev.data(), (int)c.qoff, (int)c.qsize, subj.data(), (int)c.soff, (int)c.ssize, kBlk, opt.align_size_cutoff, kTail, &ar);
                ExtTask tk;
                ext_init(tk, 0, c.qdir, (i64)rd->offset[c.qid], (i32)c.qsize, (i64)ref.offset[c.sid], (i32)c.ssize, (i32)c.qoff, (i32)c.soff);
                while (ext_plan<kBlk>(tk)) {
                    FragGeom g = ext_frag_geom(tk);
                    const int tk_qblk = tk.qblk, tk_tblk = tk.tblk; (void)tk_qblk; (void)tk_tblk;
                    MyersResult mr; HOps ops; ++n_blocks;
                    if (!tk.last && tk.qblk == kBlk && tk.tblk == kBlk) {
                        mr = run_block<kNwFull, true>(hrd->dv, href.dv, g, tk.qblk, tk.tblk, opt.error, mat, tw, R8);
                        const int done = ext_block_done(tk, mr.dist, mr.endc);
                        tail_init(ops.ts, done ? kTail : kOcaMatCnt);
                        if (mr.dist >= 0) { HMatR m{&mat}; traceback_block(tk.qblk, mr.endc + 1, m, ops); check_walk(tk.qblk, mr.endc + 1, m, ops); }
                        HRops ro{&ops.v}; HSame<kNwFull> sm{&R8, tw};
                        ext_finish_block(tk, mr.dist, mr.endc, done, ops.ts, ro, sm);
                    } else {
                        mr = run_block<kNwMax, false>(hrd->dv, href.dv, g, tk.qblk, tk.tblk, opt.error, mat, tw, R13);
                        const int done = ext_block_done(tk, mr.dist, mr.endc);
                        tail_init(ops.ts, done ? kTail : kOcaMatCnt);
                        if (mr.dist >= 0) { HMatR m{&mat}; traceback_block(tk.qblk, mr.endc + 1, m, ops); check_walk(tk.qblk, mr.endc + 1, m, ops); }
                        HRops ro{&ops.v}; HSame<kNwMax> sm{&R13, tw};
                        ext_finish_block(tk, mr.dist, mr.endc, done, ops.ts, ro, sm);
                    }
#ifdef CHECK_WINDOW
                    if (mr.dist >= 0) {
                        // the same block again, keeping only the window's records: the walk must read nothing else and give the same ops
                        static MyersRegs<kNwMax> RW;
                        for (int b = 0; b < kNwMax; ++b) { u64 lo = 0, hi = 0; if (b * 64 < tk_qblk) load64_planes(hrd->dv.bases, g.q_base, g.q_dir, g.q_comp, b * 64, &lo, &hi); RW.nlo[b] = ~lo; RW.nhi[b] = ~hi; }
                        HTgt tgw; tgw.w = tw;
                        HMatW mw; mw.init(tk_tblk, kNwMax, tk_qblk, mr.dist, mr.endc);
                        const MyersResult m2 = myers_block<kNwMax, false>(RW, tk_qblk, tk_tblk, opt.error, tgw, mw);
                        HOps o2; tail_init(o2.ts, ops.ts.M);
                        traceback_block(tk_qblk, m2.endc + 1, mw, o2);
                        ++g_win_hist[std::min(63, mw.max_per_col)];
                        g_win_stored += mw.stored; g_win_offered += mw.offered; g_win_bad += mw.bad_reads; g_win_cols += m2.endc + 1;
                        if (m2.dist != mr.dist || m2.endc != mr.endc || o2.v != ops.v) ++g_win_diff;
                        {   // and on the compact slab
                            for (int b = 0; b < kNwMax; ++b) { u64 lo = 0, hi = 0; if (b * 64 < tk_qblk) load64_planes(hrd->dv.bases, g.q_base, g.q_dir, g.q_comp, b * 64, &lo, &hi); RW.nlo[b] = ~lo; RW.nhi[b] = ~hi; }
                            HMatC mc; mc.init(tk_tblk);
                            const MyersResult m3 = myers_block<kNwMax, false>(RW, tk_qblk, tk_tblk, opt.error, tgw, mc);
                            if (mc.overflow) ++g_cmp_overflow;
                            else {
                                HOps o3; tail_init(o3.ts, ops.ts.M);
                                traceback_block(tk_qblk, m3.endc + 1, mc, o3);
                                g_cmp_bad += mc.bad_reads;
                                if (m3.dist != mr.dist || m3.endc != mr.endc || o3.v != ops.v) ++g_cmp_diff;
                            }
                        }
                    }
#endif
                    if (mr.err) { fprintf(stderr, "DP internal error %d\n", mr.err); ++bad_ext; }
                }
                double ident = tk.r_cols ? 100.0 * (double)tk.r_mat / (double)tk.r_cols : 0.0;
                int ok2 = tk.r_cols >= opt.align_size_cutoff;
                bool se = ok == ok2 && tk.r_qoff == ar.qoff && tk.r_qend == ar.qend && tk.r_toff == ar.toff && tk.r_tend == ar.tend && tk.r_cols == ar.align_size && ident == ar.ident_perc;
                if (!se) { if (bad_ext < 5) fprintf(stderr, "EXT MISMATCH read %d cand %zu: core (%d %d %d %d cols %d id %.4f) oracle (%d %d %d %d cols %d id %.4f)\n", r, ci,
                           tk.r_qoff, tk.r_qend, tk.r_toff, tk.r_tend, tk.r_cols, ident, ar.qoff, ar.qend, ar.toff, ar.tend, ar.align_size, ar.ident_perc); ++bad_ext; }
                n_m4 += ok2;
            }
        }
        ora_wfd_free(w); ora_aligner_free(al); free(oc.a);
        if (v != vid) ora_volume_free(&rd_own);
    }
#ifdef CHECK_WINDOW
    printf("window: records kept %ld of %ld offered (%.2f / %.2f words per column), reads outside the window %ld, blocks with a different walk %ld\n", g_win_stored, g_win_offered,
           g_win_cols ? (double)g_win_stored / g_win_cols : 0.0, g_win_cols ? (double)g_win_offered / g_win_cols : 0.0, g_win_bad, g_win_diff);
    printf("window: blocks by their widest column (words):");
    for (int w = 0; w < 64; ++w) if (g_win_hist[w]) printf(" %d:%ld", w, g_win_hist[w]);
    printf("\n");
    printf("compact slab of %d slots: blocks that overflow it %ld, reads outside it %ld, blocks with a different walk %ld\n", HMatC::kSlots, g_cmp_overflow, g_cmp_bad, g_cmp_diff);
    if (g_win_bad || g_win_diff || g_cmp_bad || g_cmp_diff) return 1;
#endif
    printf("check_core: candidates=%ld blocks=%ld m4=%ld seed_mismatch=%ld ext_mismatch=%ld walk_mismatch=%ld\n", n_cand, n_blocks, n_m4, bad_seed, bad_ext, g_bad_walk);
    return (bad_seed || bad_ext || g_bad_walk) ? 1 : 0;
}
