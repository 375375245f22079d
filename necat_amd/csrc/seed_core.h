// seed_core.h - per-read candidate search (k-mer sampling -> block buckets -> DDF vote -> co-linear
// gather -> chain DP -> candidate), the work one GPU lane does for one query read.
//
// Behaviour follows word_finder/word_finder.c:364 (find_candidates) and chain_dp.c:37 bit for bit;
// the data structures do not: the reference keeps a DENSE ScoringBlock table of ref_bases/b entries
// per worker (252 MB per thread for a 2 Gbp volume, word_finder.c:15-38), impossible for 10^5
// concurrent lanes, so each lane owns a sparse open-addressing table block_id -> pool slot sized from
// its own hit count, with the pool order doubling as the reference's first-touch list (blk_idx_list).
#pragma once
#include "dev_common.h"

namespace necat {

struct DevCand {          // GappedCandidate (gapped_candidate.h:9-19) with 32-bit coordinates
    i32 qid, sid, qdir, score;
    i32 qbeg, qend, qsize;
    i32 sbeg, send, ssize;
    i32 qoff, soff;
};

struct SBlock {           // ScoringBlock (word_finder_aux.h:19-25) + its blk_idx_list entry
    short score;
    short blk_offset[kBlkSeeds];
    i32 kmer_id[kBlkSeeds];
    i32 last_kmer_id;
    i32 block_id;
    i32 stale;            // blk_idx_list[].score (word_finder.c:103)
    i32 slot;             // hash slot, for O(1) reset
    i32 _pad;
};

struct SeedParams {
    int k, z, block_size, s_cutoff, align_cutoff, num_candidates, job, pairwise;
    int read_start_id, ref_start_id;
    int debug_phase;      // 0 = normal; 1 = stop after seed collection (profiling only)
    int chain_wave;       // chain DP of an evaluation on all lanes of its wave (default) or on lane 0 (NECAT_CHAIN_WAVE=0)
};

struct SeedScratch {      // all per-lane, sized from the lane's hit bound H (see seed.hip)
    u64* ht;              // [ht_mask+1] open addressing: block id (low word) | pool slot << 32, all ones = empty - one load per probe
    u32 ht_mask;
    SBlock* pool;         // [H]
    u32 pool_cap;
    u64* cs;              // chain seeds as soff<<32|qoff  [H+1]
    i32 *f, *p, *t, *v;   // [H+1] each
    u64* u;               // [H+1]
    DevCand* lcan;        // [H+1]
    u32 cs_cap;
    DevCand* out;         // candidates of this read (both strands)
    u32 out_cap;
};

constexpr int kSeedErrCapacity = -1;

NECAT_HD u32 ht_hash(i32 key, u32 mask) { return ((u32)key * 2654435761u) & mask; }

constexpr u64 kHtEmpty = ~0ULL;
NECAT_HD u64 ht_entry(i32 block_id, i32 slot) { return (u64)(u32)block_id | ((u64)(u32)slot << 32); }

NECAT_HD SBlock* sb_find(const SeedScratch& S, i32 block_id)
{
    if (block_id < 0) return nullptr;
    u32 h = ht_hash(block_id, S.ht_mask);
    for (;;) {
        const u64 e = S.ht[h];
        if ((i32)(u32)e == block_id && e != kHtEmpty) return S.pool + (u32)(e >> 32);
        if (e == kHtEmpty) return nullptr;
        h = (h + 1) & S.ht_mask;
    }
}
NECAT_HD int sb_score(const SeedScratch& S, i32 block_id)
{
    SBlock* b = sb_find(S, block_id);
    return b ? b->score : 0;
}

template <typename T>
NECAT_HD void heap_sort_u64(T* a, int n)   // ascending
{
    for (int start = n / 2 - 1; start >= 0; --start) {
        int root = start; T x = a[root];
        for (;;) { int c = 2 * root + 1; if (c >= n) break; if (c + 1 < n && a[c] < a[c + 1]) ++c; if (!(x < a[c])) break; a[root] = a[c]; root = c; }
        a[root] = x;
    }
    for (int end = n - 1; end > 0; --end) {
        T x = a[end]; a[end] = a[0];
        int root = 0;
        for (;;) { int c = 2 * root + 1; if (c >= end) break; if (c + 1 < end && a[c] < a[c + 1]) ++c; if (!(x < a[c])) break; a[root] = a[c]; root = c; }
        a[root] = x;
    }
}

// GappedCandidate_PmScoreGT (pm_worker.c:16-24): true if a sorts before b
NECAT_HD bool cand_pm_before(const DevCand& a, const DevCand& b)
{
    if (a.score != b.score) return a.score > b.score;
    if (a.qdir != b.qdir) return a.qdir < b.qdir;
    if (a.sid != b.sid) return a.sid < b.sid;
    if (a.qoff != b.qoff) return a.qoff < b.qoff;
    if (a.soff != b.soff) return a.soff < b.soff;
    return false;
}
// GappedCandidate_cdpScoreGT (chain_dp.c:26-32)
NECAT_HD bool cand_cdp_before(const DevCand& a, const DevCand& b)
{
    if (a.score != b.score) return a.score > b.score;
    if (a.qoff != b.qoff) return a.qoff < b.qoff;
    if (a.soff != b.soff) return a.soff < b.soff;
    return false;
}

template <bool PM>
NECAT_HD void sort_cands(DevCand* a, int n)   // heap sort with the comparator above (total orders)
{
    auto before = [](const DevCand& x, const DevCand& y) { return PM ? cand_pm_before(x, y) : cand_cdp_before(x, y); };
    for (int start = n / 2 - 1; start >= 0; --start) {
        int root = start; DevCand x = a[root];
        for (;;) { int c = 2 * root + 1; if (c >= n) break; if (c + 1 < n && before(a[c], a[c + 1])) ++c; if (!before(x, a[c])) break; a[root] = a[c]; root = c; }
        a[root] = x;
    }
    for (int end = n - 1; end > 0; --end) {
        DevCand x = a[end]; a[end] = a[0];
        int root = 0;
        for (;;) { int c = 2 * root + 1; if (c >= end) break; if (c + 1 < end && before(a[c], a[c + 1])) ++c; if (!before(x, a[c])) break; a[root] = a[c]; root = c; }
        a[root] = x;
    }
}

NECAT_HD int ilog2_u32(u32 v)   // chain_dp.c:18-23, v > 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 31 - __clz((int)v);
#else
    return 31 - __builtin_clz(v);
#endif
}

// word_finder.c:141-168: quotient in float, "- 1.0" and the compare in double
NECAT_HD bool ddf_ok(int dloc, int dseed, float scan_window)
{
    float q = (float)dloc / ((float)dseed * scan_window);
    double d = (double)q - 1.0;
    if (d < 0) d = -d;
    return d < 0.25;
}

// word_finder.c:141-168 is split in two so that the O(n^2) vote can be spread over the lanes of a
// wave (seed_kernels.h) while the order-dependent selection stays scalar:
//   scoring_vote_row : the votes seed i collects from / gives to the seeds j > i ("one vote per
//                      distinct later kmer_id" rule carried by tempi)
//   scoring_pick     : max vote, repeat count and the anchor choice (incl. the loc[0]==0 quirk)
template <class AddJ>
NECAT_HD int scoring_vote_row(const int* t_loc, const int* t_seedn, int i, int k, float scan_window, int read_size, AddJ& add_j)
{
    int own = 0;
    int tempi = t_seedn[i];
    for (int j = i + 1; j < k; j++)
        if (tempi != t_seedn[j] && t_seedn[j] - t_seedn[i] > 0 && t_loc[j] - t_loc[i] > 0 &&
            t_loc[j] - t_loc[i] < read_size && ddf_ok(t_loc[j] - t_loc[i], t_seedn[j] - t_seedn[i], scan_window)) {
            ++own; add_j(j); tempi = t_seedn[j];
        }
    return own;
}

NECAT_HD int scoring_pick(const int* t_loc, const int* t_seedn, const int* t_score, int* loc, int k, int* rep_loc,
                          float scan_window, int read_size)
{
    int i, j, maxval = 0, maxi = 0, rep = 0, lasti = 0;
    for (i = 0; i < k; i++) {
        if (maxval < t_score[i]) { maxval = t_score[i]; maxi = i; rep = 0; }
        else if (maxval == t_score[i]) { rep++; lasti = i; }
    }
    for (i = 0; i < 4; i++) loc[i] = 0;
    if (maxval >= 5 && rep == maxval) {
        loc[0] = t_loc[maxi]; loc[1] = t_seedn[maxi]; *rep_loc = maxi; loc[2] = t_loc[lasti]; loc[3] = t_seedn[lasti];
        return 1;
    } else if (maxval >= 5 && rep != maxval) {
        for (j = 0; j < maxi; j++)
            if (t_seedn[maxi] - t_seedn[j] > 0 && t_loc[maxi] - t_loc[j] > 0 && t_loc[maxi] - t_loc[j] < read_size &&
                ddf_ok(t_loc[maxi] - t_loc[j], t_seedn[maxi] - t_seedn[j], scan_window)) {
                if (loc[0] == 0) { loc[0] = t_loc[j]; loc[1] = t_seedn[j]; *rep_loc = j; }
                else { loc[2] = t_loc[j]; loc[3] = t_seedn[j]; }
            }
        j = maxi;
        if (loc[0] == 0) { loc[0] = t_loc[j]; loc[1] = t_seedn[j]; *rep_loc = j; }
        else { loc[2] = t_loc[j]; loc[3] = t_seedn[j]; }
        for (j = maxi + 1; j < k; j++)
            if (t_seedn[j] - t_seedn[maxi] > 0 && t_loc[j] - t_loc[maxi] > 0 && t_loc[j] - t_loc[maxi] <= read_size &&
                ddf_ok(t_loc[j] - t_loc[maxi], t_seedn[j] - t_seedn[maxi], scan_window)) {
                if (loc[0] == 0) { loc[0] = t_loc[j]; loc[1] = t_seedn[j]; *rep_loc = j; }
                else { loc[2] = t_loc[j]; loc[3] = t_seedn[j]; }
            }
        return 1;
    }
    return 0;
}

struct ScoreAdder { int* t_score; NECAT_HD void operator()(int j) { t_score[j]++; } };

NECAT_HD int scoring_seeds(const int* t_loc, const int* t_seedn, int* t_score, int* loc, int k, int* rep_loc,
                           float scan_window, int read_size)
{
    for (int i = 0; i < k; i++) t_score[i] = 0;
    ScoreAdder add; add.t_score = t_score;
    for (int i = 0; i < k - 1; i++) t_score[i] += scoring_vote_row(t_loc, t_seedn, i, k, scan_window, read_size, add);
    return scoring_pick(t_loc, t_seedn, t_score, loc, k, rep_loc, scan_window, read_size);
}

// chain_dp.c:37-159 in three steps (the wave kernel runs the first two on all lanes, seed_kernels.h).  Seeds are S.cs[0..n)
// sorted ascending by (soff, qoff); chains land in S.lcan.
constexpr int kChainMaxDist = 5000, kChainBw = 500, kChainMaxSkip = 25, kChainMinSc = 30;   // chain_dp.c:174-178

// score of seed j as predecessor of seed i (chain_dp.c:60-71), false = not a predecessor
NECAT_HD bool chain_pair_score(u64 ci, u64 cj, int kmer_size, int fj, int* sc_out)
{
    const i64 ri = (i64)(ci >> 32), qi = (i64)(ci & 0xffffffffu);
    const i64 rj = (i64)(cj >> 32), qj = (i64)(cj & 0xffffffffu);
    if (ri <= rj || qi <= qj || qi - qj > kChainMaxDist) return false;
    const i64 dr = ri - rj, dq = qi - qj;
    const i64 dd = dr > dq ? dr - dq : dq - dr;
    if (dd > kChainBw) return false;
    const i64 min_d = dq < dr ? dq : dr;
    int sc = (int)(min_d < kmer_size ? min_d : kmer_size);
    const int log_dd = dd ? ilog2_u32((u32)dd) : 0;
    sc -= (int)((double)dd * 0.01 * (double)kmer_size) + (log_dd >> 1);
    *sc_out = sc + fj;
    return true;
}

// step 1: f / p / v of every seed (chain_dp.c:46-85)
NECAT_HD void chain_fill(SeedScratch& S, int n_seeds, int kmer_size)
{
    const u64* cs = S.cs;
    i32 *f = S.f, *p = S.p, *t = S.t, *v = S.v;
    for (int i = 0; i < n_seeds; ++i) { f[i] = 0; p[i] = -1; t[i] = 0; v[i] = 0; }
    int st = 0;
    for (int i = 0; i < n_seeds; ++i) {
        const i64 ri = (i64)(cs[i] >> 32);
        int max_j = -1, max_f = kmer_size, n_skip = 0;
        while (st < i && ri - (i64)(cs[st] >> 32) > kChainMaxDist) ++st;
        for (int j = i - 1; j >= st; --j) {
            int sc;
            if (!chain_pair_score(cs[i], cs[j], kmer_size, f[j], &sc)) continue;
            if (sc > max_f) {
                max_f = sc; max_j = j;
                if (n_skip > 0) --n_skip;
            } else if (t[j] == i) {
                if (++n_skip > kChainMaxSkip) break;
            }
            if (p[j] >= 0) t[p[j]] = i;
        }
        f[i] = max_f; p[i] = max_j;
        v[i] = (max_j >= 0 && v[max_j] > max_f) ? v[max_j] : max_f;
    }
}

// the peak of the chain that ends at seed i, as the sort key of IntPair_ChainDpGT (chain_dp.c:8: first desc, second asc
// -> ascending u64 key); chain_dp.c:93-100
template <class I>
NECAT_HD u64 chain_end_key(const I* f, const I* p, const I* v, int i)
{
    int j = i;
    while (j >= 0 && f[j] < v[j]) j = p[j];
    if (j < 0) j = i;
    return ((u64)(u32)(0x7fffffff - f[j]) << 32) | (u32)j;
}

// step 2: the chain ends, best first (chain_dp.c:87-104).  Returns their number.
NECAT_HD int chain_ends(SeedScratch& S, int n_seeds)
{
    i32 *f = S.f, *p = S.p, *t = S.t, *v = S.v;
    for (int i = 0; i < n_seeds; ++i) t[i] = 0;
    for (int i = 0; i < n_seeds; ++i) if (p[i] >= 0) t[p[i]] = 1;
    int n_u = 0;
    for (int i = 0; i < n_seeds; ++i)
        if (t[i] == 0 && v[i] >= kChainMinSc) S.u[n_u++] = chain_end_key(f, p, v, i);
    if (n_u) heap_sort_u64(S.u, n_u);
    return n_u;
}

// step 3: walk the chains back, best first, every seed used once (chain_dp.c:106-159).  I = i32 (the global scratch) or
// i16 (the wave kernel's LDS copy: <= 256 seeds, scores <= 256 * k)
template <class I>
NECAT_HD int chain_emit_t(const u64* cs, const I* f, const I* p, I* t, const u64* u, DevCand* lcan,
                          int n_seeds, int n_u, int kmer_size, int min_cnt, DevCand proto)
{
    for (int i = 0; i < n_seeds; ++i) t[i] = 0;
    int n_v = 0, ncan = 0, k = 0;
    for (int i = 0; i < n_u; ++i) {
        const int n_v0 = n_v, k0 = k;
        const int first = 0x7fffffff - (int)(u32)(u[i] >> 32);
        int j = (int)(u32)(u[i] & 0xffffffffu);
        DevCand can = proto;
        can.qend = (i32)(cs[j] & 0xffffffffu) + kmer_size;
        can.send = (i32)(cs[j] >> 32) + kmer_size;
        can.qoff = can.qend; can.soff = can.send;
        int last_j = j;
        do { last_j = j; n_v++; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
        bool emit = false;
        if (j < 0) {
            if (n_v - n_v0 >= min_cnt) { can.score = first; emit = true; }
        } else if (first - f[j] >= kChainMinSc) {
            if (n_v - n_v0 >= min_cnt) { can.score = first - f[j]; emit = true; }
        }
        if (emit) {
            can.qbeg = (i32)(cs[last_j] & 0xffffffffu); can.sbeg = (i32)(cs[last_j] >> 32);
            lcan[ncan++] = can; ++k;
        }
        if (k0 == k) n_v = n_v0;
    }
    if (ncan > 1) sort_cands<false>(lcan, ncan);
    return ncan;
}

NECAT_HD int chain_emit(SeedScratch& S, int n_seeds, int n_u, int kmer_size, int min_cnt, DevCand proto)
{
    return chain_emit_t<i32>(S.cs, S.f, S.p, S.t, S.u, S.lcan, n_seeds, n_u, kmer_size, min_cnt, proto);
}

NECAT_HD int chain_dp(SeedScratch& S, int n_seeds, int kmer_size, int min_cnt, DevCand proto)
{
    chain_fill(S, n_seeds, kmer_size);
    const int n_u = chain_ends(S, n_seeds);
    if (n_u == 0) return 0;
    return chain_emit(S, n_seeds, n_u, kmer_size, min_cnt, proto);
}

// word_finder.c:171-182 (only touched blocks exist in the sparse table; zeroing an untouched block
// is a no-op in the reference as well)
NECAT_HD void clear_block_scores(const SeedScratch& S, const DevCand& can, u64 subject_start, int block_size)
{
    i64 sblk = (i64)(((u64)can.sbeg + subject_start) / (u64)block_size);
    i64 eblk = (i64)(((u64)can.send + subject_start) / (u64)block_size);
    for (i64 i = sblk; i <= eblk; ++i) { SBlock* b = sb_find(S, (i32)i); if (b) b->score = 0; }
}

// ---- find_candidate_for_one_block (word_finder.c:184-360) in stages, so that the wave kernel can
// ---- run the data-parallel ones (vote, gather) on all lanes and the rest on one lane.

// stage A: seeds of block b-1 (if it has any) followed by those of block b, offsets of b shifted by the
// block size (word_finder.c:197-222)
NECAT_HD int block_seed_lists(const SeedScratch& S, const SBlock* cur, int bs, int* kmer_id_list, int* blk_offset_list, u64* blk_start)
{
    const int block_id = cur->block_id;
    int n_seeds = 0, A = 0;
    *blk_start = (u64)bs * (u64)block_id;
    const SBlock* prev = sb_find(S, block_id - 1);
    if (prev && prev->score) {
        for (int i = 0; i < prev->score; ++i) { kmer_id_list[n_seeds] = prev->kmer_id[i]; blk_offset_list[n_seeds] = prev->blk_offset[i]; ++n_seeds; }
        A = bs;
        *blk_start = (u64)bs * (u64)(block_id - 1);
    }
    for (int i = 0; i < cur->score; ++i) { kmer_id_list[n_seeds] = cur->kmer_id[i]; blk_offset_list[n_seeds] = cur->blk_offset[i] + A; ++n_seeds; }
    return n_seeds;
}

struct AnchorGeom {       // word_finder.c:230-246
    u64 seed_tid, seed_tstart, seed_tend;
    i64 seed_tsize, stoff, seed_qoff;
    int seed_bid, bid_start, bid_end;
};

NECAT_HD AnchorGeom anchor_geometry(const DevVolume& ref, int loc0, int seedn0, u64 blk_start, int bs, int z, int qsize, i64 tid = -1)
{
    // tid >= 0: the caller already knows the subject of the anchor (the wave kernel searches seq_off on all lanes)
    AnchorGeom g;
    u64 seed_toff = (u64)loc0 + blk_start;
    g.seed_qoff = (i64)(seedn0 - 1) * z;
    g.seed_bid = (int)(seed_toff / (u64)bs);
    g.seed_tid = tid >= 0 ? (u64)tid : seq_of_offset(ref.seq_off, ref.nseq, seed_toff);
    g.seed_tstart = ref.seq_off[g.seed_tid];
    g.seed_tend = ref.seq_off[g.seed_tid + 1];
    g.seed_tsize = (i64)(g.seed_tend - g.seed_tstart);
    seed_toff -= g.seed_tstart;
    g.stoff = (i64)seed_toff;
    i64 L = g.stoff < g.seed_qoff ? g.stoff : g.seed_qoff;
    g.bid_start = g.seed_bid - (int)(L / bs) - 1;
    if (g.bid_start < 0) g.bid_start = 0;
    const i64 tr = g.seed_tsize - g.stoff, qr = (i64)qsize - g.seed_qoff;
    L = tr < qr ? tr : qr;
    g.bid_end = g.seed_bid + (int)((L + bs - 1) / bs);
    return g;
}

// stage D, one seed: is seed k of block i co-linear with the anchor (left: word_finder.c:249-275,
// right: :281-307)?  On success *key receives the chain-seed key soff<<32|qoff.
NECAT_HD bool gather_test(const AnchorGeom& g, const SBlock* sb, int k, int block_i, int bs, int z, bool right, u64* key)
{
    u64 toff = (u64)block_i * (u64)bs + (u64)(i64)sb->blk_offset[k];
    const i64 qoff = (i64)(sb->kmer_id[k] - 1) * z;
    if (!right) {
        if (toff < g.seed_tstart) return false;
        toff -= g.seed_tstart;
        if (!((i64)toff < g.stoff && qoff < g.seed_qoff)) return false;
        double s = 1.0 * (double)(u64)(g.stoff - (i64)toff) / (double)(u64)(g.seed_qoff - qoff) - 1.0;
        if (s < 0) s = -s;
        if (!(s < 0.25)) return false;
    } else {
        if (toff >= g.seed_tend) return false;
        toff -= g.seed_tstart;
        if (!((i64)toff > g.stoff && qoff > g.seed_qoff)) return false;
        double s = 1.0 * (double)(u64)((i64)toff - g.stoff) / (double)(u64)(qoff - g.seed_qoff) - 1.0;
        if (s < 0) s = -s;
        if (!(s < 0.25)) return false;
    }
    *key = (toff << 32) | (u64)(u32)qoff;
    return true;
}

// the 40 % rule of word_finder.c:273 / :305
NECAT_HD bool gather_zeroes_block(int relevant, int score) { return 1.0 * relevant / score >= 0.4; }

// stage E: sort the chain seeds, chain them, choose and emit the candidate (word_finder.c:309-358)
NECAT_HD DevCand finish_proto(const AnchorGeom& g, int qid, int qdir, int qsize)
{
    DevCand proto;
    proto.qid = qid; proto.sid = (i32)g.seed_tid; proto.qdir = qdir; proto.score = 0;
    proto.qbeg = proto.qend = 0; proto.qsize = qsize; proto.sbeg = proto.send = 0; proto.ssize = (i32)g.seed_tsize;
    proto.qoff = proto.soff = 0;
    return proto;
}

// the choice among the chains lcan[0 .. ncan) (word_finder.c:318-358).  clear_range != nullptr: the caller zeroes the blocks
// [clear_range[0], clear_range[1]] (word_finder.c:171-182) itself.
NECAT_HD int finish_choose(SeedScratch& S, const DevCand* lcan, int ncan, int seed_score, const AnchorGeom& g, const SeedParams& P,
                           int* n_out, i64* clear_range = nullptr)
{
    if (!ncan) return 0;
    const i64 seed_qoff = g.seed_qoff, stoff = g.stoff;
    auto contains = [&](const DevCand& c) {
        return seed_qoff >= c.qbeg && seed_qoff < c.qend && stoff >= c.sbeg && stoff < c.send;
    };
    DevCand can = lcan[0];
    bool emit = contains(can);
    if (!emit) {
        int max_i = ncan, max_cov = 0;
        for (int i = 0; i < ncan; ++i) {
            can = lcan[i];
            if (contains(can)) { int cov = can.qend - can.qbeg; if (cov > max_cov) { max_cov = cov; max_i = i; } }
        }
        // word_finder.c:335-343 emits `can` as the loop left it (the LAST chain), not lcanv[max_i]
        if (max_i < ncan) emit = true;
        else {
            can = lcan[0];
            if (can.qend - can.qbeg >= 5000) emit = true;
        }
    }
    if (!emit) return 0;
    can.score = seed_score; can.qoff = (i32)seed_qoff; can.soff = (i32)stoff;
    if (clear_range) {
        clear_range[0] = (i64)(((u64)can.sbeg + g.seed_tstart) / (u64)P.block_size);
        clear_range[1] = (i64)(((u64)can.send + g.seed_tstart) / (u64)P.block_size);
    } else clear_block_scores(S, can, g.seed_tstart, P.block_size);
    const bool ok = (can.send - can.sbeg >= P.align_cutoff) || (can.qend - can.qbeg >= P.align_cutoff);
    if (ok) {
        if ((u32)*n_out >= S.out_cap) return kSeedErrCapacity;
        S.out[(*n_out)++] = can;
    }
    return ok ? 1 : 0;
}

NECAT_HD int finish_candidate(SeedScratch& S, int ncs, int seed_score, const AnchorGeom& g, const SeedParams& P,
                              int qid, int qdir, int qsize, int* n_out, bool sorted = false, i64* clear_range = nullptr)
{
    if (!sorted) heap_sort_u64(S.cs, ncs);   // ChainSeedLT: (soff, qoff) ascending
    const int ncan = chain_dp(S, ncs, P.k, P.s_cutoff, finish_proto(g, qid, qdir, qsize));
    return finish_choose(S, S.lcan, ncan, seed_score, g, P, n_out, clear_range);
}

// word_finder.c:184-360, scalar composition of the stages.  Returns 1 if a candidate was appended,
// 0 if not, <0 on capacity error.
NECAT_HD int find_candidate_for_one_block(SeedScratch& S, SBlock* cur, const DevVolume& ref, const SeedParams& P,
                                          int qid, int qdir, int qsize, int* n_out)
{
    int kmer_id_list[kBlkSeeds * 2], blk_offset_list[kBlkSeeds * 2], score_list[kBlkSeeds * 2];
    const int bs = P.block_size, z = P.z;
    u64 blk_start;
    const int n_seeds = block_seed_lists(S, cur, bs, kmer_id_list, blk_offset_list, &blk_start);
    int max_score_id = -1, sc4[4];
    if (!scoring_seeds(blk_offset_list, kmer_id_list, score_list, sc4, n_seeds, &max_score_id, (float)z, qsize)) return 0;
    if (score_list[max_score_id] < 2 * P.s_cutoff) return 0;
    const AnchorGeom g = anchor_geometry(ref, sc4[0], sc4[1], blk_start, bs, z, qsize);
    int ncs = 0, seed_score = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int lo = pass == 0 ? g.bid_start : g.seed_bid, hi = pass == 0 ? g.seed_bid : g.bid_end;
        for (int i = lo; i <= hi; ++i) {
            SBlock* sb = sb_find(S, i);
            if (!sb || !sb->score) continue;
            int relevant = 0;
            for (int k = 0; k < sb->score; ++k) {
                u64 key;
                if (!gather_test(g, sb, k, i, bs, z, pass == 1, &key)) continue;
                ++relevant;
                if ((u32)ncs >= S.cs_cap) return kSeedErrCapacity;
                S.cs[ncs++] = key;
            }
            if (i != g.seed_bid && gather_zeroes_block(relevant, sb->score)) sb->score = 0;
            seed_score += relevant;
        }
        if (pass == 0) {
            if ((u32)ncs >= S.cs_cap) return kSeedErrCapacity;
            S.cs[ncs++] = ((u64)g.stoff << 32) | (u64)(u32)g.seed_qoff;
        }
    }
    return finish_candidate(S, ncs, seed_score, g, P, qid, qdir, qsize, n_out);
}

// collect_seeds (word_finder.c:107-139) incl. extract_hash_values (:66-83) and fill_one_seed (:85-104)
// for one strand of one read.  Returns the number of touched blocks (pool entries, in first-touch
// order) or kSeedErrCapacity.
NECAT_HD int seed_collect_strand(const DevVolume& ref, const IndexView& index, const u64* offset_list,
                                 const DevVolume& reads, int read_id, int qdir, const SeedParams& P, SeedScratch& S)
{
    const u64 q_goff = reads.seq_off[read_id];
    const int L = (int)(reads.seq_off[read_id + 1] - q_goff);
    u64 soff_max = ~0ULL;
    if (P.pairwise) {
        const int gid = read_id + P.read_start_id;
        if (gid >= P.ref_start_id && gid < P.ref_start_id + (int)ref.nseq) soff_max = ref.seq_off[read_id];
    }
    int nblk = 0;
    const int k = P.k, z = P.z, bs = P.block_size;
    int kmer_i = 0;
    for (int i = 0; i <= L - k; i += z, ++kmer_i) {
        // extract_hash_values (word_finder.c:66-83): first base most significant
        const u64 x = qdir == 0 ? load32_dir(reads.bases, (i64)q_goff + i, +1, 0)
                                : load32_dir(reads.bases, (i64)q_goff + L - 1 - i, -1, 1);
        const u64 hash = rev2(x) >> (64 - 2 * k);
        const u64 st = index.lookup(hash);                  // extract_kmer_list (lookup_table.c:176)
        const u64 cnt = st >> kOffsetBits;
        const u64* list = offset_list + (st & kOffsetMask);
        for (u64 kk = 0; kk < cnt; ++kk) {
            const u64 off = list[kk];
            if (off >= soff_max) continue;
            // fill_one_seed (word_finder.c:85-104)
            const i32 blk_id = (i32)(off / (u64)bs);
            const short blk_off = (short)(off % (u64)bs);
            u32 h = ht_hash(blk_id, S.ht_mask);
            SBlock* sb = nullptr;
            for (;;) {
                const u64 e = S.ht[h];
                if (e == kHtEmpty) break;
                if ((i32)(u32)e == blk_id) { sb = S.pool + (u32)(e >> 32); break; }
                h = (h + 1) & S.ht_mask;
            }
            if (!sb) {
                if ((u32)nblk >= S.pool_cap) return kSeedErrCapacity;
                sb = S.pool + nblk;
                S.ht[h] = ht_entry(blk_id, nblk);
                sb->score = 0; sb->last_kmer_id = -1; sb->block_id = blk_id; sb->stale = 0; sb->slot = (i32)h;
                ++nblk;
            }
            if (sb->last_kmer_id >= kmer_i + 1) continue;
            if (sb->score >= kBlkSeeds) continue;
            const int sid = sb->score;
            ++sb->score;
            sb->blk_offset[sid] = blk_off;
            sb->kmer_id[sid] = kmer_i + 1;
            sb->last_kmer_id = kmer_i + 1;
            sb->stale = sb->score + sb_score(S, blk_id - 1);
        }
    }
    return nblk;
}

// clear_WordFindData (word_finder.c:40-52): only touched blocks are reset
NECAT_HD void seed_reset_table(SeedScratch& S, int nblk)
{
    for (int i = 0; i < nblk; ++i) S.ht[S.pool[i].slot] = kHtEmpty;
}

// One strand of one read: word_finder.c:364-412 (find_candidates).  Candidates are appended to S.out
// with LOCAL ids.  Returns 0 or kSeedErrCapacity.
NECAT_HD int seed_one_strand(const DevVolume& ref, const IndexView& index, const u64* offset_list,
                             const DevVolume& reads, int read_id, int qdir, const SeedParams& P,
                             SeedScratch& S, int* n_out)
{
    const int L = (int)(reads.seq_off[read_id + 1] - reads.seq_off[read_id]);
    const int nblk = seed_collect_strand(ref, index, offset_list, reads, read_id, qdir, P, S);
    if (nblk < 0) return nblk;
    int rc = 0;
    for (int i = 0; i < (P.debug_phase == 1 ? 0 : nblk); ++i) {
        SBlock* sb = S.pool + i;
        if (sb->score >= P.s_cutoff && sb->stale >= 2 * P.s_cutoff) {
            int r = find_candidate_for_one_block(S, sb, ref, P, read_id, qdir, L, n_out);
            if (r < 0) { rc = r; break; }
        }
    }
    seed_reset_table(S, nblk);
    return rc;
}

// per-read post-processing of pm_search_one_volume (pm_worker.c:133-140 for job 1, :163-171 for job 0)
NECAT_HD int seed_finish_read(const SeedParams& P, SeedScratch& S, int n)
{
    if (P.job == 1 || n > P.num_candidates) {
        if (n > 1) sort_cands<true>(S.out, n);
        if (n > P.num_candidates) n = P.num_candidates;
    }
    return n;
}

// Both strands of one read, scalar.  Returns the number of candidates left in S.out (local ids), or
// <0 on capacity error.
NECAT_HD int seed_one_read(const DevVolume& ref, const IndexView& index, const u64* offset_list,
                           const DevVolume& reads, int read_id, const SeedParams& P, SeedScratch& S)
{
    int n = 0;
    int rc = seed_one_strand(ref, index, offset_list, reads, read_id, 0, P, S, &n);
    if (rc < 0) return rc;
    rc = seed_one_strand(ref, index, offset_list, reads, read_id, 1, P, S, &n);
    if (rc < 0) return rc;
    return seed_finish_read(P, S, n);
}

}  // namespace necat
