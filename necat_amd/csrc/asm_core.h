// asm_core.h - the overlapper for corrected reads, oc2asmpm (asm_pm/asmpm.c, asm_pm_common.c), restated per read:
//
//   vote      pairwise_mapping (asm_pm_common.c:509-702): every BC-th k-mer of the read (both strands) is looked up in the
//             volume's table; its hits are binned into 1000-bp blocks of the reference volume (at most 60 per block), a block with
//             enough hits (its own + its left neighbour's) is scored by find_location (:479-507: pairs of hits whose distance on the
//             reference agrees within 10 % with their distance on the read) and gives one candidate (subject read, strand, score),
//             the score topped up with the agreeing hits of the blocks to either side - which are then cleared, so the walk over the
//             blocks is order dependent and is kept in the reference's order (first touch);
//   range     compute_align_range_1 (find_mem.c:222-265): exact matches of 10-mers (every 6th of the subject against all of the
//             read), extended to maximal exact matches of >= 15, chained (km_chain.c:141-205, the minimap-style DP also used by
//             the seeding of the overlap stage); the middle match of the best chain is the anchor;
//   extend    hbn_map_extend (hbn_align.c:282-326): the block-wise aligner with 2048-bp blocks from the anchor (the clone of
//             onc_align in blockwise_edlib.c: tail match length 8, error 0.5, identity >= 65 %), then up to 300 bp of each read end
//             that it left unaligned are added by DALIGNER's local alignment (rescue.h) when that reaches the end exactly.
//
// Host code, no HIP: the cores the GPU path is built from next and the CPU model its results are checked against
// (tests/host_core/check_asmpm.cpp replays it behind the oracle's table and aligner against the reference's own oc2asmpm).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "../../include/necat_hip.h"
#include "rescue.h"

namespace necat {
namespace asmpm {

constexpr int kZV = 1000, kSM = 60, kKmerCntCutoff = 5, kBlockScoreCutoff = 4;      // asm_pm_common.c:22-33

// the reference volume as the vote sees it
struct RefView {
    const uint64_t* seq_off = nullptr;     // [nseq + 1]
    uint64_t nseq = 0;
    // lookup_table.h: hits of a k-mer, ascending
    std::function<const uint64_t*(uint64_t hash, uint64_t* n)> kmer_list;
    // packed_db.c:173-188
    int offset_to_id(int64_t offset) const
    {
        int64_t ns = (int64_t)nseq, left = 0, mid = 0, right = ns;
        while (left < right) {
            mid = (left + right) >> 1;
            if (offset >= (int64_t)seq_off[mid]) {
                if (mid == ns - 1) break;
                if (offset < (int64_t)seq_off[mid + 1]) break;
                left = mid + 1;
            } else right = mid;
        }
        return (int)mid;
    }
};

struct VoteCandidate {      // AsmGappedCandidate (asm_pm_common.c:119-125)
    int readno, score, chain, target_start, target_size, target_id, query_start;
};

// AsmGappedCandidate_ScoreGT (:145-153)
inline bool vote_before(const VoteCandidate& a, const VoteCandidate& b)
{
    if (a.score != b.score) return a.score > b.score;
    if (a.chain != b.chain) return a.chain < b.chain;
    if (a.target_id != b.target_id) return a.target_id < b.target_id;
    if (a.query_start != b.query_start) return a.query_start < b.query_start;
    return a.target_start < b.target_start;
}

// find_location (:479-507).  The ratio is computed in single precision as there (int / (int * float)).
inline bool ratio_ok(int dloc, int dseed, float len) { const float r = (float)dloc / ((float)dseed * len); return fabs((double)(r - 1.0f)) < 0.10; }
inline int find_location(const int* t_loc, const int* t_seedn, int* t_score, int* loc, int k, int* rep_loc, float len, int read_len)
{
    *rep_loc = INT32_MAX;
    int maxval = 0, maxi = 0, rep = 0, lasti = 0;
    for (int i = 0; i < k; ++i) t_score[i] = 0;
    for (int i = 0; i + 1 < k; ++i) {
        int last = t_seedn[i];
        for (int j = i + 1; j < k; ++j)
            if (last != t_seedn[j] && t_seedn[j] - t_seedn[i] > 0 && t_loc[j] - t_loc[i] > 0 && t_loc[j] - t_loc[i] < read_len &&
                ratio_ok(t_loc[j] - t_loc[i], t_seedn[j] - t_seedn[i], len)) { t_score[i]++; t_score[j]++; last = t_seedn[j]; }
    }
    for (int i = 0; i < k; ++i) {
        if (maxval < t_score[i]) { maxval = t_score[i]; maxi = i; rep = 0; }
        else if (maxval == t_score[i]) { rep++; lasti = i; }
    }
    for (int i = 0; i < 4; ++i) loc[i] = 0;
    if (maxval < 5) return 0;
    if (rep == maxval) { loc[0] = t_loc[maxi]; loc[1] = t_seedn[maxi]; *rep_loc = maxi; loc[2] = t_loc[lasti]; loc[3] = t_seedn[lasti]; return 1; }
    auto take = [&](int j) { if (loc[0] == 0) { loc[0] = t_loc[j]; loc[1] = t_seedn[j]; *rep_loc = j; } else { loc[2] = t_loc[j]; loc[3] = t_seedn[j]; } };
    for (int j = 0; j < maxi; ++j)
        if (t_seedn[maxi] - t_seedn[j] > 0 && t_loc[maxi] - t_loc[j] > 0 && t_loc[maxi] - t_loc[j] < read_len &&
            ratio_ok(t_loc[maxi] - t_loc[j], t_seedn[maxi] - t_seedn[j], len)) take(j);
    take(maxi);
    for (int j = maxi + 1; j < k; ++j)
        if (t_seedn[j] - t_seedn[maxi] > 0 && t_loc[j] - t_loc[maxi] > 0 && t_loc[j] - t_loc[maxi] <= read_len &&
            ratio_ok(t_loc[j] - t_loc[maxi], t_seedn[j] - t_seedn[maxi], len)) take(j);
    return 1;
}

struct VoteBlock {           // Back_List (:424-427)
    int score = 0, seednum = 0, index = -1;
    int loczhi[kSM], seedno[kSM];
};

struct Voter {
    std::vector<VoteBlock> db;           // one per 1000 bp of the reference volume, clean between strands
    std::vector<int> index_list, index_score;
    void init(uint64_t ref_bases) { db.assign(ref_bases / kZV + 5, VoteBlock()); index_list.clear(); index_score.clear(); }

    // one strand of one read (:545-690); soff_max: hits at or beyond it are ignored (the read's own offset when it lies in the
    // reference volume: every pair is found once, :539-543)
    void strand(const uint8_t* read, int read_len, int chain, int read_local_id, int read_global_id, int ref_start_id, const RefView& ref,
                int seed_len, int bc, int64_t soff_max, std::vector<VoteCandidate>& out)
    {
        index_list.clear(); index_score.clear();
        if (read_len < seed_len) return;          // (the reference reads past the end of such a read, asm_pm_common.c:429-446; none survive oc2mkdb's users)
        const int cleave = (read_len - seed_len) / bc + 1;
        for (int k = 0; k < cleave; ++k) {
            uint64_t h = 0;
            for (int j = 0; j < seed_len; ++j) h = (h << 2) | (uint64_t)(read[k * bc + j] & 3);
            uint64_t n = 0;
            const uint64_t* list = ref.kmer_list(h, &n);
            for (uint64_t i = 0; i < n; ++i) {
                if ((int64_t)list[i] >= soff_max) continue;
                const int b = (int)(list[i] / kZV), u = (int)(list[i] % kZV);
                VoteBlock& B = db[(size_t)b];
                if (B.score == 0 || B.seednum < k + 1) {
                    const int loc = ++B.score;
                    if (B.score > kSM) B.score = kSM;
                    if (loc <= kSM) { B.loczhi[loc - 1] = u; B.seedno[loc - 1] = k + 1; }
                    const int s = b > 0 ? B.score + db[(size_t)b - 1].score : B.score;
                    if (B.index == -1) { B.index = (int)index_list.size(); index_list.push_back(b); index_score.push_back(s); }
                    else index_score[(size_t)B.index] = s;
                }
                B.seednum = k + 1;
            }
        }
        int t_loc[2 * kSM + 30], t_seedn[2 * kSM + 30], t_score[2 * kSM + 30], loc4[4];
        const size_t touched = index_list.size();
        for (size_t i = 0; i < touched; ++i) {
            if (index_score[i] <= kKmerCntCutoff) continue;
            const int b = index_list[i];
            VoteBlock& B = db[(size_t)b];
            if (B.score == 0) continue;
            int start_loc = b * kZV, prev = 0, n = 0;
            if (b > 0) { prev = db[(size_t)b - 1].score; if (prev > 0) start_loc = (b - 1) * kZV; }
            if (prev == 0) {
                for (int j = 0; j < B.score && j < kSM; ++j) { t_loc[n] = B.loczhi[j]; t_seedn[n] = B.seedno[j]; ++n; }
            } else {
                const VoteBlock& P = db[(size_t)b - 1];
                for (int j = 0; j < prev && j < kSM; ++j) { t_loc[n] = P.loczhi[j]; t_seedn[n] = P.seedno[j]; ++n; }
                for (int j = 0; j < B.score && j < kSM; ++j) { t_loc[n] = B.loczhi[j] + kZV; t_seedn[n] = B.seedno[j]; ++n; }
            }
            int rep_loc;
            if (!find_location(t_loc, t_seedn, t_score, loc4, n, &rep_loc, (float)bc, read_len)) continue;
            if (t_score[rep_loc] < kBlockScoreCutoff) continue;
            VoteCandidate c;
            c.score = t_score[rep_loc];
            const int loc_seed = t_seedn[rep_loc];
            loc4[0] += start_loc;
            const int loc_list = loc4[0];
            const int readno = ref.offset_to_id(loc4[0]);
            const int readstart = (int)ref.seq_off[readno], length = (int)(ref.seq_off[readno + 1] - ref.seq_off[readno]), readend = readstart + length;
            if (readno + ref_start_id == read_global_id) {
                // the read itself (:604-612): its own stretch of the volume is wiped
                int u = readstart / kZV, s = readstart % kZV, k = 0;
                VoteBlock* T = &db[(size_t)u];
                for (int j = 0; j < T->score && j < kSM; ++j) if (T->loczhi[j] < s) T->loczhi[k++] = T->loczhi[j];
                T->score = k;
                for (++T, ++u, k = readend / kZV; u < k; ++u, ++T) T->score = 0;
                k = 0; s = readend % kZV;
                for (int j = 0; j < T->score && j < kSM; ++j) if (T->loczhi[j] > s) T->loczhi[k++] = T->loczhi[j];
                T->score = k;
                continue;
            }
            c.readno = read_local_id;
            loc4[1] = (loc4[1] - 1) * bc;
            c.target_id = readno; c.target_start = loc4[0] - readstart; c.target_size = length; c.query_start = loc4[1];
            const int left1 = loc4[0] - readstart + seed_len - 1, right1 = readend - loc4[0];
            const int left2 = loc4[1] + seed_len - 1, right2 = read_len - loc4[1];
            const int num1 = left1 >= left2 ? left2 : left1, num2 = right1 >= right2 ? right2 : right1;
            if (num1 + num2 < 400) continue;
            int seedcount = 0;
            // agreeing hits to the left and to the right; a block most of whose hits agree is used up (:630-647)
            for (int u = b - 2, k = num1 / kZV; u >= 0 && k >= 0; --k, --u) {
                VoteBlock& T = db[(size_t)u];
                if (T.score <= 0) continue;
                const int at = u * kZV;
                int s = 0;
                for (int j = 0; j < T.score && j < kSM; ++j)
                    if (fabs((loc_list - at - T.loczhi[j]) / ((loc_seed - T.seedno[j]) * bc * 1.0) - 1.0) < 0.10) { ++seedcount; ++s; }
                if (s * 1.0 / T.score > 0.4) T.score = 0;
            }
            for (int u = b + 1, k = num2 / kZV; k > 0; --k, ++u) {
                VoteBlock& T = db[(size_t)u];
                if (T.score <= 0) continue;
                const int at = u * kZV;
                int s = 0;
                for (int j = 0; j < T.score && j < kSM; ++j)
                    if (fabs((at + T.loczhi[j] - loc_list) / ((T.seedno[j] - loc_seed) * bc * 1.0) - 1.0) < 0.10) { ++seedcount; ++s; }
                if (s * 1.0 / T.score > 0.4) T.score = 0;
            }
            c.score += seedcount;
            c.chain = chain;
            out.push_back(c);
        }
        for (size_t i = 0; i < touched; ++i) { db[(size_t)index_list[i]].score = 0; db[(size_t)index_list[i]].index = -1; }
    }
};

// ---- range: find_mem.c, km_chain.c ---------------------------------------------------------------------------------------

struct KmerInfo { uint64_t hash; int offset, occ; };
struct Mem { int match_size, query_offset, reference_offset; };       // MaximalExactMatch (km_chain.h:11-15)

// build_kmif_list (find_mem.c:22-74) for kmer_size > window_size, as both callers have it; the first k-mer's hash is left 0 as there
inline void build_kmif(const uint8_t* s, size_t n, int kmer, int window, std::vector<KmerInfo>& out)
{
    out.clear();
    if (n < (size_t)kmer) return;
    const int stride = kmer - window;
    const uint64_t mask = (1ULL << (stride << 1)) - 1;
    uint64_t hash = 0;
    for (int j = 0; j < kmer; ++j) hash = (hash << 2) | s[j];
    out.push_back(KmerInfo{0, 0, 0});
    for (uint64_t j = (uint64_t)window; j <= n - (size_t)kmer; j += (uint64_t)window) {
        hash &= mask;
        for (int k = stride; k < kmer; ++k) hash = (hash << 2) | s[j + (uint64_t)k];
        out.push_back(KmerInfo{hash, (int)j, 0});
    }
}
// sort_kmif_list (:76-90)
// (the order is total - offsets are distinct - so any sort gives the reference's list.  build_kmif emits ascending offsets, so a STABLE sort
// by hash alone is that order: an LSD radix sort, 11 bits per pass, instead of a comparison sort - a quarter of the host time of a read)
inline void sort_kmif(std::vector<KmerInfo>& a)
{
    const size_t n = a.size();
    bool ascending = true;
    uint64_t all = 0;
    for (size_t i = 0; i < n; ++i) { all |= a[i].hash; if (i && a[i].offset <= a[i - 1].offset) ascending = false; }
    if (!ascending || n < 64) {
        std::sort(a.begin(), a.end(), [](const KmerInfo& x, const KmerInfo& y) { return x.hash < y.hash || (x.hash == y.hash && x.offset < y.offset); });
    } else {
        static thread_local std::vector<KmerInfo> tmp;
        tmp.resize(n);
        KmerInfo* src = a.data(); KmerInfo* dst = tmp.data();
        for (int shift = 0; shift < 64 && (all >> shift) != 0; shift += 11) {
            uint32_t cnt[2048 + 1] = {0};
            for (size_t i = 0; i < n; ++i) ++cnt[((src[i].hash >> shift) & 2047u) + 1];
            for (int b = 0; b < 2048; ++b) cnt[b + 1] += cnt[b];
            for (size_t i = 0; i < n; ++i) dst[cnt[(src[i].hash >> shift) & 2047u]++] = src[i];
            KmerInfo* t = src; src = dst; dst = t;
        }
        if (src != a.data()) memcpy(a.data(), src, n * sizeof(KmerInfo));
    }
    for (size_t i = 0; i < a.size();) { size_t j = i + 1; while (j < a.size() && a[j].hash == a[i].hash) ++j; a[i].occ = (int)(j - i); i = j; }
}
inline bool mem_before(const Mem& a, const Mem& b) { return a.reference_offset < b.reference_offset || (a.reference_offset == b.reference_offset && a.query_offset < b.query_offset); }

// find_kmer_match (:92-133)
inline void find_kmer_match(const std::vector<KmerInfo>& q, const std::vector<KmerInfo>& t, int kmer, int max_occ, std::vector<Mem>& out)
{
    out.clear();
    const size_t qn = q.size(), tn = t.size();
    size_t qi = 0, ti = 0;
    if (qn == 0 || tn == 0) return;
    for (;;) {
        while (qi < qn && q[qi].hash < t[ti].hash) qi += (size_t)q[qi].occ;
        if (qi >= qn) break;
        while (ti < tn && t[ti].hash < q[qi].hash) ti += (size_t)t[ti].occ;
        if (ti >= tn) break;
        if (q[qi].hash == t[ti].hash) {
            if (q[qi].occ <= max_occ && t[ti].occ <= max_occ && q[qi].occ * t[ti].occ <= max_occ)
                for (int i = 0; i < q[qi].occ; ++i) for (int j = 0; j < t[ti].occ; ++j) out.push_back(Mem{kmer, q[qi + (size_t)i].offset, t[ti + (size_t)j].offset});
            qi += (size_t)q[qi].occ; ti += (size_t)t[ti].occ;
        }
        if (qi >= qn || ti >= tn) break;
    }
}
// The same matches through a hash of the SECOND list's groups (built once per read: the read is the second sequence of every candidate's
// call, asm_pm_common.c:375-383): a probe per group of the first list instead of a merge over both - the read's list has one entry per base
// (window 1), six times the subject's.  The matches come out in another order; extend_kmer_match sorts them (mem_before is a total order).
struct KmifIndex {
    std::vector<uint64_t> key; std::vector<uint32_t> at;      // open addressing, linear probing: key = hash + 1 (0 = empty), at = first entry of the group
    uint32_t mask = 0;
    void build(const std::vector<KmerInfo>& a)
    {
        size_t groups = 0;
        for (size_t i = 0; i < a.size(); i += (size_t)a[i].occ) ++groups;
        uint32_t cap = 64;
        while (cap < 2 * groups + 2) cap <<= 1;
        mask = cap - 1;
        key.assign(cap, 0); at.assign(cap, 0);
        for (size_t i = 0; i < a.size(); i += (size_t)a[i].occ) {
            uint32_t p = (uint32_t)((a[i].hash * 0x9E3779B97F4A7C15ULL) >> 40) & mask;
            while (key[p]) p = (p + 1) & mask;
            key[p] = a[i].hash + 1; at[p] = (uint32_t)i;
        }
    }
    // index of the group of `hash` in the list the index was built from, or -1
    long find(uint64_t hash) const
    {
        uint32_t p = (uint32_t)((hash * 0x9E3779B97F4A7C15ULL) >> 40) & mask;
        for (;;) { const uint64_t k = key[p]; if (!k) return -1; if (k == hash + 1) return (long)at[p]; p = (p + 1) & mask; }
    }
};
inline void find_kmer_match_indexed(const std::vector<KmerInfo>& q, const std::vector<KmerInfo>& t, const KmifIndex& tix, int kmer, int max_occ, std::vector<Mem>& out)
{
    out.clear();
    if (q.empty() || t.empty()) return;
    for (size_t qi = 0; qi < q.size(); qi += (size_t)q[qi].occ) {
        if (q[qi].occ > max_occ) continue;
        const long ti = tix.find(q[qi].hash);
        if (ti < 0) continue;
        const int to = t[(size_t)ti].occ;
        if (to > max_occ || q[qi].occ * to > max_occ) continue;
        for (int i = 0; i < q[qi].occ; ++i) for (int j = 0; j < to; ++j) out.push_back(Mem{kmer, q[qi + (size_t)i].offset, t[(size_t)ti + (size_t)j].offset});
    }
}
// extend_kmer_match (:173-220)
inline void extend_kmer_match(std::vector<Mem>& a, const uint8_t* q, int qsize, const uint8_t* t, int tsize, int min_mem)
{
    if (a.empty()) return;
    std::sort(a.begin(), a.end(), mem_before);
    const size_t n = a.size();
    for (size_t i = 0; i < n; ++i) {
        if (a[i].match_size == 0) continue;
        int ql = a[i].query_offset, tl = a[i].reference_offset, qr = ql + a[i].match_size, tr = tl + a[i].match_size;
        while (ql && tl && q[ql - 1] == t[tl - 1]) { --ql; --tl; }
        while (qr < qsize && tr < tsize && q[qr] == t[tr]) { ++qr; ++tr; }
        a[i].query_offset = ql; a[i].reference_offset = tl; a[i].match_size = qr - ql;
        for (size_t j = i + 1; j < n && a[j].reference_offset < tr; ++j)
            if (a[j].query_offset < qr && a[j].reference_offset < tr && qr - a[j].query_offset == tr - a[j].reference_offset) a[j].match_size = 0;
        if (a[i].match_size < min_mem) a[i].match_size = 0;
    }
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) if (a[i].match_size) a[k++] = a[i];
    a.resize(k);
}

inline int ilog2_32(uint32_t v) { return v ? 31 - __builtin_clz(v) : -1; }

struct ChainRange { int qbeg, qend, qoff, sbeg, send, soff, score; };     // in the roles of the call: q = first sequence, s = second

struct Chainer {       // ChainWorkData (km_chain.c:5-23)
    int max_dist_ref = 3000, max_dist_qry = 3000, max_band_width = 500, max_skip = 25, min_cnt = 1, min_score = 100;
    std::vector<int> f, p, t, v;
    std::vector<std::pair<int, int>> u;

    // scoring_mems (:141-205)
    void score(const Mem* m, int n)
    {
        long sum = 0;
        for (int i = 0; i < n; ++i) sum += m[i].match_size;
        const int avg_cov = (int)((int)sum / n);
        f.assign((size_t)n, 0); p.assign((size_t)n, -1); t.assign((size_t)n, 0); v.assign((size_t)n, 0); u.resize((size_t)n);
        int st = 0;
        for (int i = 0; i < n; ++i) {
            const int64_t ri = m[i].reference_offset;
            const int qi = m[i].query_offset, cov = m[i].match_size;
            int max_j = -1, max_f = cov, n_skip = 0;
            while (st < i && ri > (int64_t)m[st].reference_offset + max_dist_ref) ++st;
            for (int j = i - 1; j >= st; --j) {
                if (m[j].query_offset + m[j].match_size >= qi || (int64_t)m[j].reference_offset + m[j].match_size >= ri) continue;
                const int64_t dr = ri - m[j].reference_offset;
                const int dq = qi - m[j].query_offset;
                if (dr == 0 || dq <= 0) continue;
                if (dq > max_dist_qry || dr > max_dist_ref) continue;
                const int dd = (int)(dr > dq ? dr - dq : dq - dr);
                if (dd > max_band_width) continue;
                const int min_d = (int)std::min<int64_t>(dq, dr);
                int sc = min_d > cov ? cov : min_d;
                const int log_dd = dd ? ilog2_32((uint32_t)dd) : 0;
                sc -= (int)(dd * .01 * avg_cov) + (log_dd >> 1);
                sc += f[(size_t)j];
                if (sc > max_f) { max_f = sc; max_j = j; if (n_skip) --n_skip; }
                else if (t[(size_t)j] == i) { if (++n_skip > max_skip) break; }
                if (p[(size_t)j] >= 0) t[(size_t)p[(size_t)j]] = i;
            }
            f[(size_t)i] = max_f; p[(size_t)i] = max_j;
            v[(size_t)i] = (max_j >= 0 && v[(size_t)max_j] > max_f) ? v[(size_t)max_j] : max_f;
        }
    }

    // mem_find_best_can (:322-445): the best chain's extent, score and middle match
    bool best(const Mem* m, int n, ChainRange* out)
    {
        if (n == 0) return false;
        score(m, n);
        std::fill(t.begin(), t.end(), 0);
        for (int i = 0; i < n; ++i) if (p[(size_t)i] >= 0) t[(size_t)p[(size_t)i]] = 1;
        int n_u = 0;
        for (int i = 0; i < n; ++i) {
            if (t[(size_t)i] == 0 && v[(size_t)i] >= min_score) {
                int j = i;
                while (j >= 0 && f[(size_t)j] < v[(size_t)j]) j = p[(size_t)j];
                if (j < 0) j = i;
                u[(size_t)n_u++] = std::make_pair(f[(size_t)j], j);
            }
        }
        if (n_u == 0) return false;
        std::sort(u.begin(), u.begin() + n_u);
        std::reverse(u.begin(), u.begin() + n_u);
        std::fill(t.begin(), t.end(), 0);
        int n_v = 0, k = 0;
        for (int i = 0; i < n_u; ++i) {
            const int n_v0 = n_v, k0 = k;
            int j = u[(size_t)i].second;
            do { v[(size_t)n_v++] = j; t[(size_t)j] = 1; j = p[(size_t)j]; } while (j >= 0 && t[(size_t)j] == 0);
            bool found = false;
            int sc = 0;
            if (j < 0) { if (n_v - n_v0 >= min_cnt) { sc = u[(size_t)i].first; ++k; found = true; } }
            else if (u[(size_t)i].first - f[(size_t)j] >= min_score) { if (n_v - n_v0 >= min_cnt) { sc = u[(size_t)i].first - f[(size_t)j]; ++k; found = true; } }
            if (found) {
                const Mem& first = m[(size_t)v[(size_t)n_v0]];
                const Mem& last = m[(size_t)v[(size_t)n_v - 1]];
                out->qend = first.query_offset + first.match_size; out->send = first.reference_offset + first.match_size;
                out->qbeg = last.query_offset; out->sbeg = last.reference_offset;
                out->score = sc;
                // chain_mems = the chain in ascending order; the anchor is the middle of its middle match
                const int cnt = n_v - n_v0;
                const Mem& mid = m[(size_t)v[(size_t)(n_v - 1 - cnt / 2)]];
                out->qoff = mid.query_offset + mid.match_size / 2; out->soff = mid.reference_offset + mid.match_size / 2;
                return true;
            }
            if (k0 == k) n_v = n_v0;
        }
        return false;
    }
};

// compute_align_range_1 (find_mem.c:222-265).  first / second: the two sequences in the roles the call gives them (asm_pm_common.c
// :375-383 passes the subject first and the read, with its sorted 10-mer list, second).
struct RangeFinder {
    Chainer chain;
    std::vector<KmerInfo> first_kmif;
    std::vector<Mem> mems;
    bool go(const uint8_t* first, int first_size, const uint8_t* second, int second_size, const std::vector<KmerInfo>& second_kmif, int kmer, int window,
            int min_mem, ChainRange* out, const KmifIndex* second_index = nullptr)
    {
        build_kmif(first, (size_t)first_size, kmer, window, first_kmif);
        sort_kmif(first_kmif);
        if (second_index) find_kmer_match_indexed(first_kmif, second_kmif, *second_index, kmer, 20, mems);
        else find_kmer_match(first_kmif, second_kmif, kmer, 20, mems);
        extend_kmer_match(mems, first, first_size, second, second_size, min_mem);
        std::sort(mems.begin(), mems.end(), mem_before);
        return chain.best(mems.data(), (int)mems.size(), out);
    }
};

// ---- extend: hbn_align.c ---------------------------------------------------------------------------------------------------

struct BlockAlignment { int qoff, qend, toff, tend; double ident_perc; std::string qaln, taln; };
// blockwise_edlib_align (blockwise_edlib.c:1205-1371) = onc_align with 2048-bp blocks and tail match length 8: true if the
// alignment has at least min_align_size columns (the identity test is the caller's)
using BlockAlignFn = std::function<bool(const uint8_t* read, int qoff, int qsize, const uint8_t* subject, int soff, int ssize, int min_align_size, BlockAlignment* out)>;

struct Extender {
    rescue::DalignSpec spec = rescue::spec_for_error(0.35);       // hbn_align.c:9
    rescue::Dalign dal{spec};

    // asm_pm/daligner.c:37-105 with min_align_size 1 and min_ident_perc 0.0
    bool local(const uint8_t* q, int qstart, int qsize, const uint8_t* t, int tstart, int tsize)
    {
        if (!dal.go((const char*)q, qstart, qsize, (const char*)t, tstart, tsize, 1)) return false;
        return dal.ident_perc >= 0.0;
    }

    // hbn_map_extend (:282-326); a = the block-wise alignment, updated in place
    void ends(const uint8_t* query, int qsize, const uint8_t* target, int tsize, BlockAlignment& a)
    {
        const int kMaxHang = 300, kMatchSize = 8;
        const size_t n = a.qaln.size();
        size_t from = 0, to = n;
        int left_dist = 0, right_dist = 0;
        bool lext = false, rext = false;
        int qbeg = a.qoff, tbeg = a.toff, qend = a.qend, tend = a.tend;
        {   // left_extend (:88-176)
            const int ls = std::min(qbeg, tbeg);
            if (ls <= kMaxHang && ls != 0) {
                int run = 0, qi = 0, ti = 0;
                size_t i = 0;
                for (; run < kMatchSize && i < n; ++i) {
                    const char qc = a.qaln[i], tc = a.taln[i];
                    if (qc != '-') ++qi;
                    if (tc != '-') ++ti;
                    run = qc == tc ? run + 1 : 0;
                }
                if (run >= kMatchSize) {
                    from = i + 1;
                    const int qls = qbeg + qi, tls = tbeg + ti;
                    if (local(query, qls, qls, target, tls, tls) && dal.r.aepos == qls && dal.r.bepos == tls) {
                        qbeg = dal.r.abpos; tbeg = dal.r.bbpos; left_dist = dal.r.diffs; lext = true;
                    } else from = i + 1;
                }
            }
        }
        {   // right_extend (:178-262)
            const int rs = std::min(qsize - qend, tsize - tend);
            if (rs <= kMaxHang && rs != 0) {
                int run = 0, qi = 0, ti = 0;
                size_t i = n;
                while (i && run < kMatchSize) {
                    --i;
                    const char qc = a.qaln[i], tc = a.taln[i];
                    if (qc != '-') ++qi;
                    if (tc != '-') ++ti;
                    run = qc == tc ? run + 1 : 0;
                }
                if (run >= kMatchSize) {
                    to = i;
                    const int qrs = qend - qi, trs = tend - ti;
                    if (local(query + qrs, 0, qsize - qrs, target + trs, 0, tsize - trs) && dal.r.abpos == 0 && dal.r.bbpos == 0) {
                        qend = qrs + dal.r.aepos; tend = trs + dal.r.bepos; right_dist = dal.r.diffs; rext = true;
                    }
                }
            }
        }
        if (lext || rext) {       // fix_ident_perc (:264-280)
            int diff = left_dist + right_dist;
            for (size_t i = from; i < to; ++i) if (a.qaln[i] != a.taln[i]) ++diff;
            a.ident_perc = 100.0 - 200.0 * diff / (qend + tend - qbeg - tbeg);
        }
        a.qoff = qbeg; a.toff = tbeg; a.qend = qend; a.tend = tend;
    }

    // The same on the alignment as the device returns it - n columns, two bits each (0: a base of both, 1: a query base against a gap, 2: a subject
    // base against a gap; necat_gapped_strings) - without writing the two strings out: hbn_map_extend only looks at the columns up to the first run of
    // 8 matches from either end, and fix_ident_perc's count of unequal columns between them is the alignment's total (n minus its matches, which its
    // identity 100 * matches / n gives back exactly) less those of the two ends.
    void ends_packed(const uint8_t* query, int qsize, const uint8_t* target, int tsize, const uint8_t* ops, size_t n, BlockAlignment& a)
    {
        const int kMaxHang = 300, kMatchSize = 8;
        auto op_at = [&](size_t i) { return (int)((ops[i >> 2] >> ((i & 3) * 2)) & 3); };
        size_t from = 0, to = n;
        int left_dist = 0, right_dist = 0;
        bool lext = false, rext = false;
        int qbeg = a.qoff, tbeg = a.toff, qend = a.qend, tend = a.tend;
        {
            const int ls = std::min(qbeg, tbeg);
            if (ls <= kMaxHang && ls != 0) {
                int run = 0, qi = 0, ti = 0;
                size_t i = 0;
                for (; run < kMatchSize && i < n; ++i) {
                    const int op = op_at(i);
                    const bool same = op == 0 && query[a.qoff + qi] == target[a.toff + ti];
                    if (op != 2) ++qi;
                    if (op != 1) ++ti;
                    run = same ? run + 1 : 0;
                }
                if (run >= kMatchSize) {
                    from = i + 1;
                    const int qls = qbeg + qi, tls = tbeg + ti;
                    if (local(query, qls, qls, target, tls, tls) && dal.r.aepos == qls && dal.r.bepos == tls) {
                        qbeg = dal.r.abpos; tbeg = dal.r.bbpos; left_dist = dal.r.diffs; lext = true;
                    }
                }
            }
        }
        {
            const int rs = std::min(qsize - qend, tsize - tend);
            if (rs <= kMaxHang && rs != 0) {
                int run = 0, qi = 0, ti = 0;
                size_t i = n;
                while (i && run < kMatchSize) {
                    --i;
                    const int op = op_at(i);
                    if (op != 2) ++qi;
                    if (op != 1) ++ti;
                    const bool same = op == 0 && query[a.qend - qi] == target[a.tend - ti];
                    run = same ? run + 1 : 0;
                }
                if (run >= kMatchSize) {
                    to = i;
                    const int qrs = qend - qi, trs = tend - ti;
                    if (local(query + qrs, 0, qsize - qrs, target + trs, 0, tsize - trs) && dal.r.abpos == 0 && dal.r.bbpos == 0) {
                        qend = qrs + dal.r.aepos; tend = trs + dal.r.bepos; right_dist = dal.r.diffs; rext = true;
                    }
                }
            }
        }
        if (lext || rext) {
            int diff = left_dist + right_dist;
            if (from < to) {
                const long matches = llround(a.ident_perc * (double)n / 100.0);
                long unequal = (long)n - matches;
                int q = a.qoff, t = a.toff;
                for (size_t i = 0; i < from; ++i) { const int op = op_at(i); if (!(op == 0 && query[q] == target[t])) --unequal; q += op != 2; t += op != 1; }
                q = a.qend; t = a.tend;
                for (size_t i = n; i > to; --i) { const int op = op_at(i - 1); q -= op != 2; t -= op != 1; if (!(op == 0 && query[q] == target[t])) --unequal; }
                diff += (int)unequal;
            }
            a.ident_perc = 100.0 - 200.0 * diff / (qend + tend - qbeg - tbeg);
        }
        a.qoff = qbeg; a.toff = tbeg; a.qend = qend; a.tend = tend;
    }
};

// ---- one read: extend_candidates (asm_pm_common.c:329-422) ------------------------------------------------------------------

struct ReadMapper {
    RangeFinder range;
    Extender ext;
    std::vector<KmerInfo> read_kmif;
    std::vector<uint8_t> subject;
    BlockAlignment aln;

    // cands: the votes of both strands; fwd_read: the read's forward strand; subject_of(sid, strand, out): a reference read.
    // Appends the read's records (ids local: the caller adds the start ids; REV subjects already turned to forward coordinates).
    void go(std::vector<VoteCandidate>& cands, int num_extended, const uint8_t* fwd_read, int read_id, int read_size,
            const std::function<void(int sid, int strand, std::vector<uint8_t>& out)>& subject_of, const BlockAlignFn& block_align, std::vector<necat_m4>& out)
    {
        if (cands.empty()) return;
        std::sort(cands.begin(), cands.end(), vote_before);
        build_kmif(fwd_read, (size_t)read_size, 10, 1, read_kmif);
        sort_kmif(read_kmif);
        const size_t first = out.size();
        for (size_t i = 0; i < cands.size() && (int)i < num_extended; ++i) {
            const VoteCandidate& vc = cands[i];
            const int sid = vc.target_id, sdir = vc.chain;          // normalise_candidate_qoff (:133-143): a REV query = a REV subject
            bool seen = false;
            for (size_t j = first; j < out.size() && !seen; ++j) seen = out[j].sid == sid && out[j].sdir == sdir;      // :155-169 (qdir is always FWD)
            if (seen) continue;
            subject_of(sid, sdir, subject);
            const int ssize = (int)subject.size();
            ChainRange r;
            if (!range.go(subject.data(), ssize, fwd_read, read_size, read_kmif, 10, 6, 15, &r)) continue;
            // the roles come back exchanged (:386-394): the chain's first sequence is the subject
            const int qoff = r.soff, soff = r.qoff;
            if (!block_align(fwd_read, qoff, read_size, subject.data(), soff, ssize, 400, &aln)) continue;
            if (!(aln.ident_perc >= 65.0)) continue;
            ext.ends(fwd_read, read_size, subject.data(), ssize, aln);
            necat_m4 m;
            memset(&m, 0, sizeof m);
            m.qid = read_id; m.sid = sid; m.ident_perc = aln.ident_perc; m.vscore = r.score; m.qdir = 0;
            m.qoff = (uint64_t)aln.qoff; m.qend = (uint64_t)aln.qend; m.qext = (uint64_t)qoff; m.qsize = (uint64_t)read_size;
            m.sdir = sdir; m.soff = (uint64_t)aln.toff; m.send = (uint64_t)aln.tend; m.sext = (uint64_t)soff; m.ssize = (uint64_t)ssize;
            out.push_back(m);
        }
        for (size_t j = first; j < out.size(); ++j) {       // :401-415
            necat_m4& m = out[j];
            if (m.sdir == 1) { const uint64_t so = m.ssize - m.send, se = m.ssize - m.soff; m.soff = so; m.send = se; }
        }
    }
};

// The same walk in two halves, for a block aligner that takes all reads' anchors at once (the device: necat_asm_align_batch).  What a candidate's
// range and alignment come out as depends only on (read, subject, strand), so a later candidate of the same subject and strand is either skipped -
// a record exists - or fails exactly as the first one did: one anchor per (subject, strand), in the walk's order, gives the same records.
struct Planned { int sid, sdir, qoff, soff, score, ssize; };

struct BatchMapper {
    RangeFinder range;
    Extender ext;
    std::vector<KmerInfo> read_kmif;
    KmifIndex read_index;
    std::vector<uint8_t> subject;

    void plan(std::vector<VoteCandidate>& cands, int num_extended, const uint8_t* fwd_read, int read_size,
              const std::function<void(int sid, int strand, std::vector<uint8_t>& out)>& subject_of, std::vector<Planned>& out)
    {
        if (cands.empty()) return;
        std::sort(cands.begin(), cands.end(), vote_before);
        build_kmif(fwd_read, (size_t)read_size, 10, 1, read_kmif);
        sort_kmif(read_kmif);
        read_index.build(read_kmif);
        const size_t first = out.size();
        for (size_t i = 0; i < cands.size() && (int)i < num_extended; ++i) {
            const int sid = cands[i].target_id, sdir = cands[i].chain;
            bool seen = false;
            for (size_t j = first; j < out.size() && !seen; ++j) seen = out[j].sid == sid && out[j].sdir == sdir;
            if (seen) continue;
            subject_of(sid, sdir, subject);
            ChainRange r;
            Planned p;
            p.sid = sid; p.sdir = sdir; p.ssize = (int)subject.size(); p.qoff = p.soff = -1; p.score = 0;        // qoff < 0: no chain, nothing to align
            if (range.go(subject.data(), p.ssize, fwd_read, read_size, read_kmif, 10, 6, 15, &r, &read_index)) { p.qoff = r.soff; p.soff = r.qoff; p.score = r.score; }
            out.push_back(p);
        }
    }

    // the same with the alignments' columns as the device packed them (ops[k], a[k].qaln / taln unused): Extender::ends_packed
    void finish_packed(const Planned* planned, size_t n, const bool* ok, BlockAlignment* a, const uint8_t* const* ops, const size_t* ncols, const uint8_t* fwd_read, int read_id,
                       int read_size, const std::function<void(int sid, int strand, std::vector<uint8_t>& out)>& subject_of, std::vector<necat_m4>& out)
    {
        for (size_t k = 0; k < n; ++k) {
            const Planned& p = planned[k];
            if (p.qoff < 0 || !ok[k] || !(a[k].ident_perc >= 65.0)) continue;
            subject_of(p.sid, p.sdir, subject);
            ext.ends_packed(fwd_read, read_size, subject.data(), p.ssize, ops[k], ncols[k], a[k]);
            necat_m4 m;
            memset(&m, 0, sizeof m);
            m.qid = read_id; m.sid = p.sid; m.ident_perc = a[k].ident_perc; m.vscore = p.score; m.qdir = 0;
            m.qoff = (uint64_t)a[k].qoff; m.qend = (uint64_t)a[k].qend; m.qext = (uint64_t)p.qoff; m.qsize = (uint64_t)read_size;
            m.sdir = p.sdir; m.soff = (uint64_t)a[k].toff; m.send = (uint64_t)a[k].tend; m.sext = (uint64_t)p.soff; m.ssize = (uint64_t)p.ssize;
            if (m.sdir == 1) { const uint64_t so = m.ssize - m.send, se = m.ssize - m.soff; m.soff = so; m.send = se; }
            out.push_back(m);
        }
    }

    // a[k]: the block-wise alignment of planned[k] (ok = at least 400 columns), strings filled
    void finish(const Planned* planned, size_t n, const bool* ok, BlockAlignment* a, const uint8_t* fwd_read, int read_id, int read_size,
                const std::function<void(int sid, int strand, std::vector<uint8_t>& out)>& subject_of, std::vector<necat_m4>& out)
    {
        for (size_t k = 0; k < n; ++k) {
            const Planned& p = planned[k];
            if (p.qoff < 0 || !ok[k] || !(a[k].ident_perc >= 65.0)) continue;
            subject_of(p.sid, p.sdir, subject);
            ext.ends(fwd_read, read_size, subject.data(), p.ssize, a[k]);
            necat_m4 m;
            memset(&m, 0, sizeof m);
            m.qid = read_id; m.sid = p.sid; m.ident_perc = a[k].ident_perc; m.vscore = p.score; m.qdir = 0;
            m.qoff = (uint64_t)a[k].qoff; m.qend = (uint64_t)a[k].qend; m.qext = (uint64_t)p.qoff; m.qsize = (uint64_t)read_size;
            m.sdir = p.sdir; m.soff = (uint64_t)a[k].toff; m.send = (uint64_t)a[k].tend; m.sext = (uint64_t)p.soff; m.ssize = (uint64_t)p.ssize;
            if (m.sdir == 1) { const uint64_t so = m.ssize - m.send, se = m.ssize - m.soff; m.soff = so; m.send = se; }
            out.push_back(m);
        }
    }
};

}  // namespace asmpm
}  // namespace necat
