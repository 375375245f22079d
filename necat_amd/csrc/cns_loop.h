// cns_loop.h - host side of necat_cns_extension_batch: the consensus stage's extension loop
// (consensus/consensus_one_read.c:221-372, consensus/error_estimate.c:96-183) run for many templates at once.
//
// The reference walks one template's candidates in score order and decides, one candidate at a time,
// whether to align it (read not used yet, region not yet covered max_cov deep) and whether to keep the
// alignment; every decision depends on the alignments kept before it.  Here the alignments are the expensive
// part and live on the device, so the loop is split in two:
//
//   select  per template, the next candidates the loop WOULD align if none of the pending alignments changed
//           its state (a prefix of the walk, at most `spec` candidates) - all templates'
//           selections form one device batch;
//   replay  with the results in hand the walk is repeated in order with the reference's rules; a selected
//           candidate the sequential loop would have skipped after all (its region got covered by an overlap
//           accepted a moment earlier, or the 15 identities were complete) is simply not used.
//
// Both conditions that make the loop skip a candidate (read already used, region fully covered) only ever
// turn from false to true, so a candidate skipped at selection time is skipped by the sequential loop too:
// the replay sees exactly the alignments it needs, and the overlaps, their order and the per-template
// numbers are the sequential ones.
//
// No HIP in this header: the alignments come from a callback (the device path in stage_cns.inl; the CPU test
// tests/host_core/check_cns.cpp plugs the oracle's aligner in to check this logic without a GPU).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/necat_hip.h"

namespace necat {
namespace cns {

// rules of the loop -----------------------------------------------------------------------------------

// consensus/consensus_aux.c:92-113
inline bool full_cov_ovlp(int ql, int qr, int qs, int tl, int tr, int ts, int min_len, int tail)
{
    const bool q_left = ql <= tail, q_right = qs - qr <= tail, t_left = tl <= tail, t_right = ts - tr <= tail;
    if ((q_left && q_right) || (t_left && t_right)) return true;
    if (q_right) { if (!t_left) return false; if (qr - ql >= min_len) return true; }
    if (t_right) { if (!q_left) return false; if (qr - ql >= min_len) return true; }
    return false;
}
// consensus/consensus_aux.c:115-122
inline bool mapping_range_ok(int ql, int qr, int qs, int tl, int tr, int ts, int min_len, double ratio)
{
    return qr - ql >= min_len || tr - tl >= min_len || qr - ql >= qs * ratio || tr - tl >= ts * ratio;
}
// consensus/error_estimate.c:7-29: the overlap reaches an end of a read on both sides (200 bp slack)
inline bool end_to_end(int qoff, int qend, int qsize, int toff, int tend, int tsize)
{
    const int slack = 200;
    const bool ql = qoff <= slack, qr = qsize - qend <= slack, tl = toff <= slack, tr = tsize - tend <= slack;
    return (ql && qr) || (tl && tr) || (qr && tl) || (tr && ql);
}
// consensus/consensus_one_read.c:11-16
inline double overlap_weight(double ident_perc)
{
    const double e = (100.0 - ident_perc) / 100.0 / 2.0;
    double w = (1.0 - e) * (1.0 - e) + e * e / 3.0;
    if (100.0 - ident_perc <= 1.0e-6) w = 1.0;
    return w;
}
// consensus/error_estimate.c:31-63 (same order of floating-point operations)
inline double ident_lower_bound(const double* ident, int n)
{
    if (n < 5) return 0.0;
    double sum = 0.0;
    for (int i = 0; i < n; ++i) sum += ident[i];
    const double avg = sum / n;
    double se = 0.0;
    for (int i = 0; i < n; ++i) se += (avg - ident[i]) * (avg - ident[i]);
    se /= n;
    se = sqrt(se);
    return avg - se * 5;
}

constexpr int kEstimateCandidates = 50;   // error_estimate.c:120
constexpr int kIdentSamples = 15;         // error_estimate.c:115
constexpr int kGroup = 50;                // consensus_one_read.c:321

// one computed alignment as the loop sees it
struct Aligned {
    necat_alignment a;
    uint32_t block;       // where its columns are
    uint64_t off;
};

// Aligns `n` candidates; fills out[0..n).  Returns 0 or a NECAT_ERR code.
using AlignFn = std::function<int(const necat_candidate* cands, uint64_t n, Aligned* out)>;

// The two loops over a template's coverage array, compiled for AVX2 as well (picked at load time): they are most
// of the replay's time.  (Host code; the attribute is dropped in hipcc's device pass over this header.)
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define NECAT_HOST_SIMD __attribute__((target_clones("avx2", "default")))
#else
#define NECAT_HOST_SIMD
#endif
NECAT_HOST_SIMD
inline bool all_covered(const uint16_t* c, int n, uint16_t m)
{
    int i = 0;
    for (; i + 64 <= n; i += 64) {                  // branch-free inside a chunk so that the comparison vectorises
        unsigned below = 0;
        for (int k = 0; k < 64; ++k) below |= (unsigned)(c[i + k] < m);
        if (below) return false;
    }
    for (; i < n; ++i) if (c[i] < m) return false;
    return true;
}
NECAT_HOST_SIMD
inline void add_one(uint16_t* c, int n) { for (int i = 0; i < n; ++i) ++c[i]; }

// A fixed-capacity array inside one of the call-wide arenas of run(): the per-template lists (selection, used
// reads, pooled / accepted overlaps, ranges) are bounded by the template's candidate count, and 23 000 templates
// growing four std::vectors each cost more in malloc and first-touch page faults than the loop's own arithmetic.
template <class T>
struct Span {
    T* p = nullptr;
    uint32_t n = 0;
    uint32_t size() const { return n; }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    T* begin() { return p; }
    T* end() { return p + n; }
    const T* begin() const { return p; }
    const T* end() const { return p + n; }
    const T* data() const { return p; }
    void push_back(const T& v) { p[n++] = v; }
    void clear() { n = 0; }
};

struct Template {
    // input
    const necat_candidate* c = nullptr;   // its candidates in examination order
    uint64_t c_base = 0;                  // index of c[0] in the caller's array
    uint32_t n = 0, n_all = 0;
    int tsize = 0;
    // loop state
    enum Stage : uint8_t { ESTIMATE, COVER, DONE } stage = DONE;
    bool examined = false;
    uint32_t cursor = 0;        // next candidate the walk looks at
    uint32_t group_end = 0;     // COVER: end of the current group of 50 (0 = a new group starts at cursor)
    uint32_t stop = 0;          // where this round's selection stopped (exclusive)
    Span<uint32_t> sel;         // this round's selection (candidate indices, ascending); capacity min(n, 50)
    uint64_t sel_at = 0;        // index of sel[0] in the round's batch
    Span<int32_t> used;         // reads already extended for this template (ReadIdPool); capacity n
    Span<uint16_t> cov;         // coverage of the template by accepted overlaps (tsize + 1 entries, zeroed by the arena)
    Span<necat_cns_overlap> pool;   // OverlapsPool of the estimate stage: lives in the slot of `overlaps` (which is empty
                                    // until the stage ends) and is compacted into it in place
    double ident[kIdentSamples];
    int n_ident = 0;
    // output
    double ident_cutoff = 0.0;
    int num_can = 0, num_ovlps = 0;
    Span<necat_cns_overlap> overlaps;   // capacity n (a candidate is accepted at most once)
    Span<int32_t> ranges;               // capacity 2 x min(n, 50): only the estimate stage's overlaps add ranges
    uint64_t n_used = 0;
    uint64_t covered_bp = 0;     // sum of the accepted overlaps' target ranges
    uint32_t est_aligned = 0;    // alignments consumed by the estimate stage
    uint32_t cover_passes = 0;   // device passes this template took part in during the cover stage

    bool is_used(int32_t qid) const { return std::find(used.begin(), used.end(), qid) != used.end(); }
    // consensus_one_read.c:145-151
    bool region_full(int from, int to, int max_cov) const { return from >= to || all_covered(cov.p + from, to - from, (uint16_t)max_cov); }
    void cover(int from, int to) { if (to > from) add_one(cov.p + from, to - from); }
    static necat_cns_overlap record(uint64_t cand, const Aligned& al)
    {
        necat_cns_overlap o;
        o.cand = cand; o.qoff = al.a.qoff; o.qend = al.a.qend; o.toff = al.a.toff; o.tend = al.a.tend;
        o.align_size = al.a.align_size; o.ops_block = al.block; o.ops_off = al.off;
        o.ident_perc = al.a.ident_perc; o.weight = 0.0;
        return o;
    }
    void accept(necat_cns_overlap o)
    {
        o.weight = overlap_weight(o.ident_perc);
        overlaps.push_back(o);
        ++num_ovlps;
        cover(o.toff, o.tend);
        covered_bp += (uint64_t)(o.tend - o.toff);
    }
};

struct Knobs {
    int spec_estimate_extra = 1;   // estimate stage: selected = identities still missing (x alignments per identity so far) + this;
                                   // < 0: strict mode for tests - only the identities still missing, one read once per
                                   //      pass, no widening: with spec_cover = 1 no alignment is ever computed in vain
    int spec_cover = 12;           // cover stage: candidates selected per round; 0 = from the coverage still missing:
    double adapt_mult = 1.25;      //   max(adapt_min, 2 + adapt_mult * missing coverage / mean accepted overlap length)
    int adapt_min = 2;
};

// ---- select -------------------------------------------------------------------------------------------

inline void finish_estimate(Template& t, uint32_t next, const necat_cns_options& opt);

// true if `qid` is the read of a candidate already selected this round
inline bool in_selection(const Template& t, int32_t qid)
{
    for (uint32_t s : t.sel) if (t.c[s].qid == qid) return true;
    return false;
}

inline void select(Template& t, const necat_cns_options& opt, const Knobs& kn)
{
    const bool strict = kn.spec_estimate_extra < 0;      // test mode: never select what the loop might not reach
    bool have_prefix = false;
    t.sel.clear();
    while (t.stage != Template::DONE) {
        if (t.stage == Template::ESTIMATE) {
            const uint32_t limit = std::min<uint32_t>(t.n, kEstimateCandidates);
            // identities still missing, scaled by how many alignments it took per identity so far
            const uint32_t need = (uint32_t)(kIdentSamples - t.n_ident);
            const uint32_t want = kn.spec_estimate_extra < 0 ? need :      // exactly what the loop aligns whatever the results
                std::min<uint32_t>(kEstimateCandidates,
                (t.n_ident > 0 ? (need * t.est_aligned + t.n_ident - 1) / t.n_ident : need * (t.est_aligned ? 2u : 1u)) + (uint32_t)kn.spec_estimate_extra);
            uint32_t i = t.cursor;
            for (; i < limit && t.sel.size() < want; ++i) {
                if (t.is_used(t.c[i].qid)) continue;
                if (strict && in_selection(t, t.c[i].qid)) break;
                t.sel.push_back(i);                       // a second candidate of a selected read is selected too: whether
            }                                             // the loop reaches it depends on the first one's alignment
            t.stop = i;
            if (!t.sel.empty()) return;
            finish_estimate(t, limit, opt);                // ran out of candidates: error_estimate.c:178 with i = limit
            continue;
        }
        // COVER: consensus_one_read.c:317-372.  The coverage does not change during a selection: one prefix count of
        // the positions still below max_cov answers every "is this region full" of the walk in O(1).
        static thread_local std::vector<uint32_t> below;
        if (!have_prefix) {
            below.resize((size_t)t.tsize + 1);
            uint32_t acc = 0;
            for (int x = 0; x < t.tsize; ++x) { below[x] = acc; acc += t.cov[x] < opt.max_cov; }
            below[t.tsize] = acc;
            have_prefix = true;
        }
        auto full = [&](int from, int to) { return from >= to || below[to] == below[from]; };
        if (t.group_end == 0) {
            if (t.cursor >= t.n || full(0, t.tsize)) { t.stage = Template::DONE; break; }
            t.group_end = std::min<uint32_t>(t.cursor + kGroup, t.n);
        }
        // the few templates that need a third, fourth ... pass select twice as many each time: a pass of a handful of
        // alignments costs as much as one of ten thousand
        int want = strict ? kn.spec_cover : std::min(kGroup, kn.spec_cover << std::min<uint32_t>(t.cover_passes, 3));
        if (kn.spec_cover <= 0) {
            // as many as the missing coverage asks for, at the mean length of the overlaps accepted so far (+ 25 %)
            uint64_t missing = 0;
            for (int x = 0; x < t.tsize; ++x) missing += t.cov[x] < opt.max_cov ? (uint64_t)(opt.max_cov - t.cov[x]) : 0;
            const double mean = t.num_ovlps ? (double)t.covered_bp / t.num_ovlps : 0.6 * t.tsize;
            want = (int)std::min<double>(kGroup, std::max<double>(kn.adapt_min, 2.0 + kn.adapt_mult * (double)missing / std::max(1.0, mean)));
        }
        uint32_t i = t.cursor;
        for (; i < t.group_end && (int)t.sel.size() < want; ++i) {
            const necat_candidate& c = t.c[i];
            if (t.is_used(c.qid)) continue;
            if (full((int)c.sbeg, (int)c.send)) continue;
            if (strict && in_selection(t, c.qid)) break;
            t.sel.push_back(i);
        }
        t.stop = i;
        if (!t.sel.empty()) { ++t.cover_passes; return; }
        t.cursor = i;                                      // nothing to align up to here
        if (t.cursor == t.group_end) t.group_end = 0;
    }
}

// ---- replay -------------------------------------------------------------------------------------------

// get_idents + the cutoff (error_estimate.c:65-94, :180-183), add_extended_overlaps (consensus_one_read.c:153-190)
inline void finish_estimate(Template& t, uint32_t next, const necat_cns_options& opt)
{
    auto qsize_of = [&](const necat_cns_overlap& p) { return (int)t.c[p.cand - t.c_base].qsize; };
    if (t.n_ident < kIdentSamples) {
        int k = 0;
        for (const auto& p : t.pool) {
            if (k == kIdentSamples) break;
            if (end_to_end(p.qoff, p.qend, qsize_of(p), p.toff, p.tend, t.tsize)) t.ident[k++] = p.ident_perc;
        }
        if (k < kIdentSamples) {
            k = 0;
            for (const auto& p : t.pool) {
                if (k == kIdentSamples) break;
                if (p.qend - p.qoff >= qsize_of(p) * 0.6 || p.tend - p.toff >= t.tsize * 0.6) t.ident[k++] = p.ident_perc;
            }
        }
        t.n_ident = k;
    }
    std::sort(t.ident, t.ident + t.n_ident, [](double a, double b) { return a > b; });
    int n = t.n_ident;
    if (n >= 8) n = (int)(n * 0.7);
    t.ident_cutoff = ident_lower_bound(t.ident, n);
    for (uint32_t j = 0; j < t.pool.size(); ++j) {
        const necat_cns_overlap p = t.pool[j];            // by value: accept() writes into the same slot, at or before j
        const int qs = qsize_of(p);
        if (p.ident_perc < t.ident_cutoff) continue;
        if (!mapping_range_ok(p.qoff, p.qend, qs, p.toff, p.tend, t.tsize, opt.min_align_size, opt.mapping_ratio)) continue;
        t.accept(p);
        if (full_cov_ovlp(p.qoff, p.qend, qs, p.toff, p.tend, t.tsize, 1000, 200)) { t.ranges.push_back(p.toff); t.ranges.push_back(p.tend); }
    }
    t.pool.clear();
    t.num_can = (int)next;                                 // consensus_one_read.c:312-313
    t.cursor = next; t.group_end = 0;
    t.stage = Template::COVER;
}

// `res` = the alignments of t.sel, in that order
inline void replay(Template& t, const Aligned* res, const necat_cns_options& opt)
{
    if (t.sel.empty()) return;
    size_t k = 0;
    if (t.stage == Template::ESTIMATE) {
        for (uint32_t i = t.cursor; i < t.stop; ++i) {
            const necat_candidate& c = t.c[i];
            const bool selected = k < t.sel.size() && t.sel[k] == i;
            const Aligned* alp = selected ? &res[k++] : nullptr;
            if (t.is_used(c.qid)) continue;               // at selection time already, or by an earlier candidate of this pass
            const Aligned& al = *alp;                      // candidates that pass the test were all selected
            ++t.n_used; ++t.est_aligned;
            if (!al.a.ok) continue;
            t.pool.push_back(Template::record(t.c_base + i, al));
            t.used.push_back(c.qid);
            if (end_to_end(al.a.qoff, al.a.qend, (int)c.qsize, al.a.toff, al.a.tend, t.tsize)) {
                t.ident[t.n_ident++] = al.a.ident_perc;
                if (t.n_ident == kIdentSamples) { finish_estimate(t, i, opt); return; }   // error_estimate.c:172-178: i stays
            }
        }
        t.cursor = t.stop;
        if (t.cursor >= std::min<uint32_t>(t.n, kEstimateCandidates)) finish_estimate(t, t.cursor, opt);
        return;
    }
    for (uint32_t i = t.cursor; i < t.stop; ++i) {
        const necat_candidate& c = t.c[i];
        const bool selected = k < t.sel.size() && t.sel[k] == i;
        if (!selected) continue;                           // skipped at selection time: still skipped
        const Aligned& al = res[k++];
        if (t.is_used(c.qid)) continue;                    // a candidate of the same read was accepted earlier in this pass
        if (t.region_full((int)c.sbeg, (int)c.send, opt.max_cov)) continue;   // covered in the meantime: not aligned by the loop
        ++t.n_used;
        ++t.num_can;
        if (!al.a.ok) continue;
        const necat_alignment& a = al.a;
        if (a.ident_perc < t.ident_cutoff && !full_cov_ovlp(a.qoff, a.qend, (int)c.qsize, a.toff, a.tend, t.tsize, 5000, 100)) continue;
        if (!mapping_range_ok(a.qoff, a.qend, (int)c.qsize, a.toff, a.tend, t.tsize, opt.min_align_size, opt.mapping_ratio)) continue;
        t.accept(Template::record(t.c_base + i, al));
        t.used.push_back(c.qid);
    }
    t.cursor = t.stop;
    if (t.cursor == t.group_end) t.group_end = 0;
}

// ---- driver -------------------------------------------------------------------------------------------

template <class F>
inline void parallel_for(size_t n, F&& fn)
{
    static const unsigned cap = []() { const char* e = getenv("NECAT_CNS_THREADS"); const int v = e ? atoi(e) : 0; return v > 0 ? (unsigned)v : 32u; }();
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    nt = (unsigned)std::min<size_t>(std::min<unsigned>(nt, cap), (n + 63) / 64);
    if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    std::atomic<size_t> next(0);
    std::vector<std::thread> th;
    auto work = [&]() { for (;;) { const size_t b = next.fetch_add(64); if (b >= n) break; for (size_t i = b; i < std::min(n, b + 64); ++i) fn(i); } };
    for (unsigned t = 0; t + 1 < nt; ++t) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
}

struct Stats {
    uint64_t n_aligned = 0, n_used = 0; uint32_t n_rounds = 0; double init_ms = 0, select_ms = 0, gather_ms = 0, replay_ms = 0;
    std::unique_ptr<necat_cns_overlap[]> ov_arena;     // Template::overlaps / ::ranges of a finished run() point into these
    std::unique_ptr<int32_t[]> rg_arena;
};

inline double now_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

// Runs the loop of every template to its end.  Returns 0 or the callback's error.
// buffers a caller may keep between calls (the library keeps one per context)
struct Scratch {
    uint16_t* cov = nullptr;
    size_t cov_cap = 0;
    Scratch() = default;
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    ~Scratch() { free(cov); }
};

inline int run(std::vector<Template>& ts, const necat_cns_options& opt, const Knobs& kn, const AlignFn& align, Stats* st, Scratch* sc = nullptr)
{
    double t0 = now_ms();
    // call-wide arenas for the per-template lists
    std::vector<uint64_t> at_n(ts.size() + 1, 0), at_50(ts.size() + 1, 0);
    for (size_t i = 0; i < ts.size(); ++i) { at_n[i + 1] = at_n[i] + ts[i].n; at_50[i + 1] = at_50[i] + std::min<uint32_t>(ts[i].n, kGroup); }
    std::unique_ptr<necat_cns_overlap[]> ov_arena(new necat_cns_overlap[at_n.back() + 1]);
    std::unique_ptr<int32_t[]> used_arena(new int32_t[at_n.back() + 1]), rg_arena(new int32_t[2 * at_50.back() + 1]);
    std::unique_ptr<uint32_t[]> sel_arena(new uint32_t[at_50.back() + 1]);
    // coverage arrays: one allocation kept between calls (first-touch page faults of a fresh 0.4 GB block cost more than
    // the whole replay: 168 ms from 32 threads at once), zeroed per template in the parallel loop below
    std::vector<uint64_t> at_cov(ts.size() + 1, 0);
    for (size_t i = 0; i < ts.size(); ++i) at_cov[i + 1] = at_cov[i] + (uint64_t)ts[i].tsize + 1;
    Scratch local;
    if (!sc) sc = &local;
    if (at_cov.back() + 1 > sc->cov_cap) {
        free(sc->cov);
        sc->cov_cap = (at_cov.back() + 1) + (at_cov.back() + 1) / 4;
        sc->cov = (uint16_t*)malloc(sc->cov_cap * sizeof(uint16_t));
        if (!sc->cov) { sc->cov_cap = 0; return NECAT_ERR_MEMORY; }
    }
    uint16_t* const cov_arena = sc->cov;
    st->ov_arena = std::move(ov_arena); st->rg_arena = std::move(rg_arena);       // the results point into these two
    parallel_for(ts.size(), [&](size_t i) {
        Template& t = ts[i];
        t.overlaps.p = st->ov_arena.get() + at_n[i]; t.pool.p = t.overlaps.p;
        t.used.p = used_arena.get() + at_n[i];
        t.ranges.p = st->rg_arena.get() + 2 * at_50[i];
        t.sel.p = sel_arena.get() + at_50[i];
        t.examined = t.n > 0 && (uint32_t)opt.min_cov <= t.n_all;      // consensus_one_read.c:223
        if (!t.examined) { t.stage = Template::DONE; return; }
        t.cov.p = cov_arena + at_cov[i]; t.cov.n = (uint32_t)t.tsize + 1;
        memset(t.cov.p, 0, (size_t)t.cov.n * sizeof(uint16_t));
        if (opt.use_fixed_ident_cutoff) {                               // :267-272
            t.ident_cutoff = 100.0 * (1.0 - opt.error);
            t.stage = Template::COVER;
        } else {
            t.stage = Template::ESTIMATE;
        }
    });
    std::vector<necat_candidate> batch;
    std::vector<Aligned> res;
    st->init_ms += now_ms() - t0;
    for (;;) {
        t0 = now_ms();
        parallel_for(ts.size(), [&](size_t i) { if (ts[i].stage != Template::DONE) select(ts[i], opt, kn); else ts[i].sel.clear(); });
        st->select_ms += now_ms() - t0; t0 = now_ms();
        uint64_t total = 0;
        for (auto& t : ts) { t.sel_at = total; total += t.sel.size(); }
        if (total == 0) break;
        batch.resize(total); res.resize(total);
        parallel_for(ts.size(), [&](size_t i) {
            Template& t = ts[i];
            for (size_t k = 0; k < t.sel.size(); ++k) batch[t.sel_at + k] = t.c[t.sel[k]];
        });
        st->gather_ms += now_ms() - t0;
        const int rc = align(batch.data(), total, res.data());
        if (rc) return rc;
        t0 = now_ms();
        parallel_for(ts.size(), [&](size_t i) { if (!ts[i].sel.empty()) replay(ts[i], res.data() + ts[i].sel_at, opt); });
        st->replay_ms += now_ms() - t0;
        st->n_aligned += total; ++st->n_rounds;
    }
    for (auto& t : ts) { st->n_used += t.n_used; t.cov.p = nullptr; t.cov.n = 0; }
    return 0;
}

// ---- candidate partitions (the records between oc2pmov -j 0 / oc2pcan and oc2cns) ----------------------

struct Packed { uint32_t w[7]; };          // PackedGappedCandidate, common/gapped_candidate.h:64-85
constexpr uint32_t kSdirBit = 1u << 31, kQdirBit = 1u << 30, kOffBit = 1u << 29, kScoreMask = kOffBit - 1;
constexpr uint32_t kMaxExamined = 300;     // MAX_EXAMINED_CAN, consensus/consensus_aux.h:15

// common/gapped_candidate.c:71-93: the same pair seen with the subject on its forward strand
inline void normalise_sdir(Packed& p, uint32_t qsize, uint32_t ssize)
{
    if (!(p.w[0] & kSdirBit)) return;
    p.w[0] = (p.w[0] & kScoreMask) | ((p.w[0] & kQdirBit) ? 0 : kQdirBit) | ((p.w[0] & kOffBit) ? 0 : kOffBit);
    const uint32_t qb = qsize - p.w[6], qe = qsize - p.w[5], sb = ssize - p.w[3], se = ssize - p.w[2];
    p.w[5] = qb; p.w[6] = qe; p.w[2] = sb; p.w[3] = se;
}

// PackedGappedCandidate_CnsScoreGT (gapped_candidate.c:95-121): score down, then qid, qdir, qbeg, sbeg up;
// records equal under it are ordered by their remaining words (the reference leaves them to its introsort)
inline bool examined_before(const Packed& a, const Packed& b)
{
    const int sa = (int)(a.w[0] & kScoreMask), sb = (int)(b.w[0] & kScoreMask);
    if (sa != sb) return sa > sb;
    if (a.w[4] != b.w[4]) return (int)a.w[4] < (int)b.w[4];
    const int da = (a.w[0] & kQdirBit) != 0, db = (b.w[0] & kQdirBit) != 0;
    if (da != db) return da < db;
    if (a.w[5] != b.w[5]) return (int)a.w[5] < (int)b.w[5];
    if (a.w[2] != b.w[2]) return (int)a.w[2] < (int)b.w[2];
    for (int k = 0; k < 7; ++k) if (a.w[k] != b.w[k]) return a.w[k] < b.w[k];
    return false;
}

// common/gapped_candidate.c:31-52
inline necat_candidate unpack(const Packed& p)
{
    necat_candidate c;
    memset(&c, 0, sizeof c);
    c.sdir = (p.w[0] & kSdirBit) ? 1 : 0; c.qdir = (p.w[0] & kQdirBit) ? 1 : 0; c.score = (int32_t)(p.w[0] & kScoreMask);
    c.sid = (int32_t)p.w[1]; c.sbeg = p.w[2]; c.send = p.w[3];
    c.qid = (int32_t)p.w[4]; c.qbeg = p.w[5]; c.qend = p.w[6];
    if (p.w[0] & kOffBit) { c.qoff = c.qbeg; c.soff = c.sbeg; } else { c.qoff = c.qend; c.soff = c.send; }
    return c;
}

// consensus_one_partition.c:10-96 + consensus_one_read.c:250-260.  seq_off = the read set's prefix offsets.
// Returns the index of the first bad record + 1, or 0.  Records are bucketed by template with a counting sort,
// the (small) buckets are ordered and unpacked in parallel.
inline uint64_t load_partition(std::vector<Packed>& recs, const uint64_t* seq_off, uint64_t nseq,
                               std::vector<necat_candidate>& cands, std::vector<uint64_t>& off, std::vector<uint64_t>& n_all)
{
    const uint64_t n = recs.size();
    std::atomic<uint64_t> bad(0);
    parallel_for(n, [&](size_t i) {
        Packed& p = recs[i];
        bool ok = p.w[1] < nseq && p.w[4] < nseq;
        if (ok) {
            const uint64_t qsize = seq_off[p.w[4] + 1] - seq_off[p.w[4]], ssize = seq_off[p.w[1] + 1] - seq_off[p.w[1]];
            ok = p.w[5] <= p.w[6] && p.w[6] <= qsize && p.w[2] <= p.w[3] && p.w[3] <= ssize;
            if (ok) normalise_sdir(p, (uint32_t)qsize, (uint32_t)ssize);
        }
        if (!ok) { uint64_t cur = bad.load(); while ((cur == 0 || i + 1 < cur) && !bad.compare_exchange_weak(cur, i + 1)) {} }
    });
    if (bad.load()) return bad.load();
    // bucket by template id
    std::vector<uint64_t> start(nseq + 1, 0);
    for (uint64_t i = 0; i < n; ++i) ++start[recs[i].w[1] + 1];
    std::vector<uint32_t> tmpl;                       // templates that have candidates, ascending
    for (uint64_t s = 0; s < nseq; ++s) { if (start[s + 1]) tmpl.push_back((uint32_t)s); start[s + 1] += start[s]; }
    std::vector<Packed> sorted(n);
    {
        std::vector<uint64_t> at(start.begin(), start.end() - 1);
        for (uint64_t i = 0; i < n; ++i) sorted[at[recs[i].w[1]]++] = recs[i];
    }
    const size_t nt = tmpl.size();
    off.assign(nt + 1, 0); n_all.assign(nt, 0);
    for (size_t t = 0; t < nt; ++t) {
        const uint64_t cnt = start[tmpl[t] + 1] - start[tmpl[t]];
        n_all[t] = cnt;
        off[t + 1] = off[t] + std::min<uint64_t>(cnt, kMaxExamined);
    }
    cands.resize(off[nt]);
    parallel_for(nt, [&](size_t t) {
        Packed* b = sorted.data() + start[tmpl[t]];
        const uint64_t cnt = n_all[t], keep = off[t + 1] - off[t];
        std::sort(b, b + cnt, examined_before);
        for (uint64_t k = 0; k < keep; ++k) {
            necat_candidate c = unpack(b[k]);
            c.qsize = seq_off[c.qid + 1] - seq_off[c.qid]; c.ssize = seq_off[c.sid + 1] - seq_off[c.sid];
            cands[off[t] + k] = c;
        }
    });
    recs.swap(sorted);
    return 0;
}

}  // namespace cns
}  // namespace necat
