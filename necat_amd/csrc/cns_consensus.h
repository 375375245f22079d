// cns_consensus.h - the consensus proper of NECAT's oc2cns (tasc/: alignment tags -> backbone -> best path), host side.
//
// Input: what necat_cns_extension_batch hands over per template - the accepted overlaps in add_one_align order, each with
// its weight, its target range and its alignment columns (2 bits per column, include/necat_hip.h) - plus the bases of
// the reads.  Output: the corrected segments exactly as the reference prints them (consensus_broken / consensus_unbroken,
// tasc/cbcns.c:108-264).  Pure host code: one template is a few hundred thousand tags, sorted and walked once; templates
// are independent and run on all host threads while the GPU aligns the next partition.
//
// Bit-exactness notes (reference paths relative to /root/reference/src/):
//   * a tag is (t_pos, delta, q_base | p_t_pos, p_delta, p_q_base) + the overlap's weight (tasc/align_tags.c:22-71).  Tags
//     that agree in all six keys differ only in weight, and the link weight is the sum of their weights IN ARRAY ORDER
//     (tasc/cns_aux.c:46-48) - a sum of doubles, so the order klib's (unstable) introsort leaves equal tags in is part of
//     the result.  klib_introsort below is that algorithm (klib/ksort.h:180-232: median-of-three quicksort on an
//     explicit stack, ranges of <= 16 left for one final insertion sort, comb sort when the depth budget runs out),
//     so the permutation is the reference's.
//   * scores are doubles compared with > in a fixed visiting order (tasc/cns_aux.c:150-183); kept as written.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

namespace necat_host {
namespace cns {

// tasc/align_tags.h:8-16.  The six keys are kept as two words in AlignTag_LT's order (tasc/align_tags.c:6-18: t_pos, delta, q_base, then p_t_pos,
// p_delta, p_q_base; positions biased so that -1 sorts first), so that the sort - two thirds of the consensus proper's time - compares two integers
// instead of walking six fields; the comparisons' outcomes, and with them the permutation klib's introsort leaves, are the same.
struct Tag {                // 16 bytes: the keys, and in the 16 low bits of `lo` (not part of the order) which overlap of the template the tag comes from - its weight is looked up
    uint64_t hi, lo;
    static uint64_t key(int pos, uint8_t delta, char base) { return (uint64_t)((uint32_t)pos ^ 0x80000000u) << 32 | (uint64_t)delta << 24 | (uint64_t)(uint8_t)base << 16; }
    void set(int t_pos_, uint8_t delta_, char q_base_, int p_t_pos_, uint8_t p_delta_, char p_q_base_, uint32_t ovl) { hi = key(t_pos_, delta_, q_base_); lo = key(p_t_pos_, p_delta_, p_q_base_) | (ovl & 0xffffu); }
    int t_pos() const { return (int)((uint32_t)(hi >> 32) ^ 0x80000000u); }
    int p_t_pos() const { return (int)((uint32_t)(lo >> 32) ^ 0x80000000u); }
    uint8_t delta() const { return (uint8_t)(hi >> 24); }
    uint8_t p_delta() const { return (uint8_t)(lo >> 24); }
    char q_base() const { return (char)(uint8_t)(hi >> 16); }
    char p_q_base() const { return (char)(uint8_t)(lo >> 16); }
    uint32_t ovl() const { return (uint32_t)(lo & 0xffffu); }
};
constexpr size_t kMaxTagOverlaps = 65536;       // overlaps of one template (MAX_EXAMINED_CAN = 300, consensus_aux.h:15)

struct TagLess { bool operator()(const Tag& a, const Tag& b) const { typedef unsigned __int128 u128; return ((u128)a.hi << 64 | (a.lo & ~0xffffULL)) < ((u128)b.hi << 64 | (b.lo & ~0xffffULL)); } };
inline bool tag_less(const Tag& a, const Tag& b) { return TagLess()(a, b); }

// ---- klib's introsort, same decisions in the same order (see the header comment) ----
template <class T, class Less>
void klib_insertsort(T* s, T* t, Less lt)
{
    for (T* i = s + 1; i < t; ++i)
        for (T* j = i; j > s && lt(*j, *(j - 1)); --j) { T x = *j; *j = *(j - 1); *(j - 1) = x; }
}

template <class T, class Less>
void klib_combsort(size_t n, T* a, Less lt)
{
    const double shrink = 1.2473309501039786540366528676643;
    size_t gap = n;
    bool swapped;
    do {
        if (gap > 2) { gap = (size_t)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
        swapped = false;
        for (T* i = a; i < a + n - gap; ++i) {
            T* j = i + gap;
            if (lt(*j, *i)) { T x = *i; *i = *j; *j = x; swapped = true; }
        }
    } while (swapped || gap > 2);
    if (gap != 1) klib_insertsort(a, a + n, lt);
}

template <class T, class Less>
void klib_introsort(size_t n, T* a, Less lt)
{
    if (n < 1) return;
    if (n == 2) { if (lt(a[1], a[0])) { T x = a[0]; a[0] = a[1]; a[1] = x; } return; }
    int d = 2;
    while ((1ul << d) < n) ++d;
    struct Frame { T* left; T* right; int depth; };
    std::vector<Frame> stack;
    stack.reserve(sizeof(size_t) * d + 2);
    T* s = a; T* t = a + (n - 1);
    d <<= 1;
    for (;;) {
        if (s < t) {
            if (--d == 0) { klib_combsort((size_t)(t - s + 1), s, lt); t = s; continue; }
            T* i = s; T* j = t; T* k = i + ((j - i) >> 1) + 1;
            if (lt(*k, *i)) { if (lt(*k, *j)) k = j; }
            else k = lt(*j, *i) ? i : j;
            const T rp = *k;
            if (k != t) { T x = *k; *k = *t; *t = x; }
            for (;;) {
                do ++i; while (lt(*i, rp));
                do --j; while (i <= j && lt(rp, *j));
                if (j <= i) break;
                T x = *i; *i = *j; *j = x;
            }
            { T x = *i; *i = *t; *t = x; }
            if (i - s > t - i) {
                if (i - s > 16) stack.push_back(Frame{s, i - 1, d});
                s = t - i > 16 ? i + 1 : t;
            } else {
                if (t - i > 16) stack.push_back(Frame{i + 1, t, d});
                t = i - s > 16 ? i - 1 : s;
            }
        } else {
            if (stack.empty()) { klib_insertsort(a, a + n, lt); return; }
            const Frame f = stack.back(); stack.pop_back();
            s = f.left; t = f.right; d = f.depth;
        }
    }
}

// ---- tags of one overlap (get_cns_tags, tasc/align_tags.c:22-71) ----
// ops: the alignment's columns, 2 bits each (0 match, 1 query base over '-', 2 '-' over target base, 3 mismatch);
// qbase(i): byte code 0..3 of base i of the query STRAND, starting at the alignment's qoff.
// Returns false (no tags added) when a run of >= 255 query bases sits between two target bases (:38-42).
template <class QBase>
bool overlap_tags(const uint8_t* ops, int ncols, QBase qbase, int toff, uint32_t ovl, std::vector<Tag>& tags)
{
    auto op_at = [&](int i) { return (ops[i >> 2] >> ((i & 3) * 2)) & 3; };
    int jj = 0;
    for (int i = 0; i < ncols; ++i) {
        const int op = op_at(i);
        if (op != 2) ++jj;               // qaln[i] != '-'
        if (op != 1) jj = 0;             // taln[i] != '-'
        if (jj >= 255) return false;
    }
    static const char dec[4] = {'A', 'C', 'G', 'T'};
    Tag tag;
    jj = 0;
    int j = toff - 1, p_j = -1, p_jj = 0, qi = 0;
    char p_q = '-';
    const size_t at = tags.size();
    tags.resize(at + (size_t)ncols);
    Tag* out = tags.data() + at;
    for (int i = 0; i < ncols; ++i) {
        const int op = op_at(i);
        char q = '-';
        if (op != 2) { q = dec[qbase(qi) & 3]; ++qi; ++jj; }
        if (op != 1) { ++j; jj = 0; }
        tag.set(j, (uint8_t)jj, q, p_j, (uint8_t)p_jj, p_q, ovl);
        p_j = j; p_jj = jj; p_q = q;
        out[i] = tag;
    }
    return true;
}

// ---- backbone (tasc/cns_aux.c:22-125) ----
struct Link { double weight; int p_t_pos; uint8_t p_delta; char p_q_base; int count; };          // LinkInfo
struct BaseLinks { int n_link = 0, coverage = 0; uint32_t first = 0; int best_p_t_pos = -1; uint8_t best_p_delta = 255, best_p_q_base = '.'; double score = 0; };
struct DeltaCov { BaseLinks links[5]; };                                                            // DeltaCovInfo
struct Item { int n_delta = 0; uint32_t first = 0; };                                               // BackboneItem

inline int base_code(char c)           // encode_dna_base, tasc/cns_aux.c:7-20
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}

struct Backbone {
    std::vector<Item> items;          // [template_size]
    std::vector<DeltaCov> deltas;     // the DeltaCovInfo allocator
    std::vector<Link> links;          // the LinkInfo allocator
    std::vector<int> coverage;        // [template_size]

    // tags: all overlaps' tags of the template; sorted in place.  weight[tag.ovl()]: the weight of the overlap a tag comes from
    void build(std::vector<Tag>& tags, const double* weight, int template_size)
    {
        items.assign((size_t)template_size, Item());
        coverage.assign((size_t)template_size, 0);
        deltas.clear(); links.clear();
        const int ntag = (int)tags.size();
        klib_introsort((size_t)ntag, tags.data(), TagLess());
        const Tag* T = tags.data();
        int i = 0;
        while (i < ntag) {                                               // one target position (build_backbone :110-124)
            int j = i + 1;
            while (j < ntag && T[i].t_pos() == T[j].t_pos()) ++j;
            Item& it = items[(size_t)T[i].t_pos()];
            it.n_delta = T[j - 1].delta() + 1;
            it.first = (uint32_t)deltas.size();
            deltas.resize(deltas.size() + (size_t)it.n_delta);           // value-initialised: a delta without tags keeps coverage 0
            int a = i;
            while (a < j) {                                              // one delta (build_backbone_item :84-94)
                int b = a + 1;
                while (b < j && T[a].delta() == T[b].delta()) ++b;
                DeltaCov& dc = deltas[it.first + T[a].delta()];
                for (int q = 0; q < 5; ++q) dc.links[q] = BaseLinks();
                int c = a;
                while (c < b) {                                          // one query base (build_delta_links :66-73)
                    int e = c;
                    while (e < b && T[c].q_base() == T[e].q_base()) ++e;
                    BaseLinks& bl = dc.links[base_code(T[c].q_base())];
                    bl.coverage = e - c;
                    bl.first = (uint32_t)links.size();
                    int n_link = 0;
                    int g = c;
                    while (g < e) {                                      // one predecessor (build_base_links :36-51)
                        int h = g + 1;
                        while (h < e && T[g].p_t_pos() == T[h].p_t_pos() && T[g].p_delta() == T[h].p_delta() && T[g].p_q_base() == T[h].p_q_base()) ++h;
                        Link L; L.p_t_pos = T[g].p_t_pos(); L.p_delta = T[g].p_delta(); L.p_q_base = T[g].p_q_base(); L.count = h - g; L.weight = 0;
                        for (int k = g; k < h; ++k) L.weight += weight[T[k].ovl()];
                        links.push_back(L);
                        ++n_link;
                        g = h;
                    }
                    bl.n_link = n_link;
                    c = e;
                }
                if (T[a].delta() == 0) coverage[(size_t)T[a].t_pos()] = b - a;
                a = b;
            }
            i = j;
        }
    }

    // consensus_backbone_segment (tasc/cns_aux.c:127-217): best-scoring path through [from, to); out = base codes 0..3
    void segment(int from, int to, std::string& out, int* cns_from, int* cns_to)
    {
        BaseLinks* g_best = nullptr;
        int g_best_t_pos = 0, g_best_q = -1;
        double g_best_score = -1.0;
        for (int i = from; i < to; ++i) {
            const Item& it = items[(size_t)i];
            for (int j = 0; j < it.n_delta; ++j) {
                for (int kk = 0; kk < 5; ++kk) {
                    BaseLinks& col = deltas[it.first + j].links[kk];
                    if (!col.coverage) continue;
                    double best_score = -1;
                    for (int ck = 0; ck < col.n_link; ++ck) {
                        const Link& L = links[col.first + ck];
                        const int pi = L.p_t_pos, pj = L.p_delta, pkk = base_code(L.p_q_base);
                        double score = L.weight - 0.4 * 0.5 * coverage[(size_t)i];
                        if (pi != -1) score += deltas[items[(size_t)pi].first + pj].links[pkk].score;
                        if (score > best_score) {
                            best_score = score;
                            col.best_p_t_pos = pi; col.best_p_delta = (uint8_t)pj; col.best_p_q_base = (uint8_t)pkk;
                        }
                    }
                    col.score = best_score;
                    if (best_score > g_best_score) { g_best_score = best_score; g_best = &col; g_best_t_pos = i; g_best_q = kk; }
                }
            }
        }
        out.clear();
        int cfrom = 0;
        const int cto = g_best_t_pos + 1;
        if (g_best) {
            int ck = g_best_q;
            for (;;) {
                const int bb = ck;
                const int i = g_best->best_p_t_pos;
                if (i == -1) break;
                const int j = g_best->best_p_delta;
                ck = g_best->best_p_q_base;
                g_best = &deltas[items[(size_t)i].first + j].links[ck];
                cfrom = i;
                if (bb != 4) out.push_back((char)bb);
            }
            for (size_t a = 0, b = out.size(); a + 1 < b; ++a, --b) { const char x = out[a]; out[a] = out[b - 1]; out[b - 1] = x; }
        }
        if (cns_from) *cns_from = cfrom;
        if (cns_to) *cns_to = cto;
    }
};

struct Segment { int left, right; std::string seq; };      // one record of cns_out / raw_out: target range + bases as letters

// consensus_broken (tasc/cbcns.c:108-163): every stretch covered >= min_cov deep and long enough gets its own consensus
inline void consensus_broken(Backbone& bb, int min_cov, int min_size, int template_size, std::vector<Segment>& cns)
{
    int i = 0;
    const int* cov = bb.coverage.data();
    std::string seq;
    while (i < template_size) {
        while (i < template_size && cov[i] < min_cov) ++i;
        int j = i + 1;
        while (j < template_size && cov[j] >= min_cov) ++j;
        if (j - i >= min_size * 0.85) {
            bb.segment(i, j, seq, nullptr, nullptr);
            if ((int)seq.size() >= min_size) {
                static const char dec[4] = {'A', 'C', 'G', 'T'};
                Segment sg; sg.left = i; sg.right = j; sg.seq = seq;
                for (char& c : sg.seq) c = dec[(int)c & 3];
                cns.push_back(std::move(sg));
            }
        }
        i = j;
    }
}

// consensus_unbroken (tasc/cbcns.c:171-264): corrected stretches stitched together with the raw bases between them.
// raw(k) = byte code of template base k.  Returns the number of corrected stretches; out = letters.
template <class Raw>
int consensus_unbroken(Backbone& bb, int min_cov, int min_size, Raw raw, int template_size, std::string& out)
{
    struct Iv { int raw_from, raw_to, cns_from, cns_to; };
    std::vector<Iv> ivs;
    std::string all, frag;
    const int* cov = bb.coverage.data();
    int i = 0;
    out.clear();
    while (i < template_size) {
        while (i < template_size && cov[i] < min_cov) ++i;
        int j = i + 1;
        while (j < template_size && cov[j] >= min_cov) ++j;
        if (j - i >= min_size * 0.85) {
            int rf = 0, rt = 0;
            bb.segment(i, j, frag, &rf, &rt);
            if ((int)frag.size() >= min_size) {
                Iv v; v.raw_from = rf; v.raw_to = rt; v.cns_from = (int)all.size(); v.cns_to = v.cns_from + (int)frag.size();
                all += frag;
                ivs.push_back(v);
            }
        }
        i = j;
    }
    if (ivs.empty()) return 0;
    int last_raw_to = 0;
    for (const Iv& v : ivs) {
        for (int k = last_raw_to; k < v.raw_from; ++k) out.push_back((char)raw(k));
        out.append(all, (size_t)v.cns_from, (size_t)(v.cns_to - v.cns_from));
        last_raw_to = v.raw_to;
    }
    for (int k = last_raw_to; k < template_size; ++k) out.push_back((char)raw(k));
    static const char dec[4] = {'A', 'C', 'G', 'T'};
    for (char& c : out) c = dec[(int)c & 3];
    return (int)ivs.size();
}

// get_raw_intvs (consensus/consensus_one_read.c:19-67): the stretches of the read no consensus covers, >= 1000 bases
inline void raw_intervals(int read_size, const std::vector<Segment>& cns, std::vector<std::pair<int, int>>& raw)
{
    bool first = true, last_open = true;
    int left = 0, right;
    const int end_offset = read_size - 1;
    for (const Segment& s : cns) {
        const int fs = s.left, fe = s.right;
        if (first) {
            last_open = false; first = false;
            if (fs > 0) left = 0;
            else { left = fe + 1; continue; }
        }
        right = fs - 1;
        if (right - left + 1 >= 1000) raw.emplace_back(left, right);
        if (fe >= end_offset) { last_open = false; break; }
        left = fe + 1;
    }
    if (last_open) {
        right = end_offset;
        if (right - left + 1 >= 1000) raw.emplace_back(left, right);
    }
}

// ---- one template, from its overlaps to its output records ----
inline bool is_ontsa_hdr(const char* h) { return strncmp(h, "ontsa_id", 8) == 0; }      // common/cns_seq.c:11-22

// DUMP_CNS_SEQ (common/cns_seq.h:24-44)
inline void append_record(std::string& out, const char* hdr, int id, int left, int right, const std::string& seq, int org_size, int num_can, int num_ovlps,
                          double ident_cutoff)
{
    char buf[512];
    if (is_ontsa_hdr(hdr)) { out += '>'; out += hdr; out += '_'; }
    else { snprintf(buf, sizeof buf, ">ontsa_id_%d_", id); out += buf; }
    snprintf(buf, sizeof buf, "(%d_%d_%d_%d_%d_%d_%lf)\n", left, right, (int)seq.size(), org_size, num_can, num_ovlps, ident_cutoff);
    out += buf;
    out += seq;
    out += '\n';
}

struct OverlapIn {            // one add_one_align call (tasc/cbcns.c:47)
    const uint8_t* ops;       // columns, 2 bits each
    int ncols;
    int toff;                 // first target base of the alignment
    double weight;
    const uint8_t* qfwd;      // the query read, forward strand, byte codes
    int qsize, qoff, qdir;    // strand (qdir 1: reverse complement) and the alignment's start on it
};

struct Worker {               // per-thread scratch
    Backbone bb;
    std::vector<Tag> tags;
    std::vector<double> weights;      // of the template's overlaps, by Tag::ovl()
    std::vector<Segment> segs;
    std::vector<std::pair<int, int>> raw;
    std::string seq;
};

// What consensus_one_read does after its extension loop (consensus/consensus_one_read.c:373-395): returns whether the
// template counts as corrected; appends its records to cns_txt / raw_txt.
inline bool consensus_template(Worker& w, const OverlapIn* ov, size_t n_ov, const uint8_t* tseq, int tsize, int tid, const char* hdr,
                               int min_cov, int min_size, bool full_consensus, int num_can, int num_ovlps, double ident_cutoff,
                               std::string& cns_txt, std::string& raw_txt)
{
    w.tags.clear(); w.weights.resize(n_ov);
    if (n_ov > kMaxTagOverlaps) { fprintf(stderr, "[cns] template %d: %zu overlaps exceed the tag format\n", tid, n_ov); abort(); }
    for (size_t k = 0; k < n_ov; ++k) {
        const OverlapIn& o = ov[k];
        const uint8_t* q = o.qfwd;
        const int qsize = o.qsize, qoff = o.qoff;
        w.weights[k] = o.weight;
        if (o.qdir == 0) overlap_tags(o.ops, o.ncols, [&](int i) { return q[qoff + i]; }, o.toff, (uint32_t)k, w.tags);
        else overlap_tags(o.ops, o.ncols, [&](int i) { return (uint8_t)(3 - q[qsize - 1 - (qoff + i)]); }, o.toff, (uint32_t)k, w.tags);
    }
    w.bb.build(w.tags, w.weights.data(), tsize);
    static const char dec[4] = {'A', 'C', 'G', 'T'};
    if (full_consensus) {
        const int n = consensus_unbroken(w.bb, min_cov, min_size, [&](int k) { return tseq[k]; }, tsize, w.seq);
        if (n) append_record(cns_txt, hdr, tid, 0, tsize, w.seq, tsize, num_can, num_ovlps, ident_cutoff);
        return n != 0;
    }
    w.segs.clear(); w.raw.clear();
    consensus_broken(w.bb, min_cov, min_size, tsize, w.segs);
    for (const Segment& sg : w.segs) append_record(cns_txt, hdr, tid, sg.left, sg.right, sg.seq, tsize, num_can, num_ovlps, ident_cutoff);
    raw_intervals(tsize, w.segs, w.raw);
    for (const auto& iv : w.raw) {
        const int from = iv.first, to = iv.second + 1;
        w.seq.resize((size_t)(to - from));
        for (int k = from; k < to; ++k) w.seq[(size_t)(k - from)] = dec[tseq[k] & 3];
        append_record(raw_txt, hdr, tid, from, to, w.seq, tsize, num_can, num_ovlps, ident_cutoff);
    }
    return true;
}

// a read nobody corrected goes out whole (consensus_one_partition.c:172-194)
inline void uncorrected_record(std::string& out, const uint8_t* b, int size, int id, const char* hdr)
{
    static const char dec[4] = {'A', 'C', 'G', 'T'};
    std::string seq((size_t)size, 'A');
    for (int k = 0; k < size; ++k) seq[(size_t)k] = dec[b[k] & 3];
    append_record(out, hdr, id, 0, size, seq, size, 0, 0, 0.0);
}

}  // namespace cns
}  // namespace necat_host
