// rescue.h - the "rescue" pair the reference falls back on when the block-wise extension stops short of a candidate's chain
// (consensus_aux.c:170-195 with -r 1, rm_worker.c:104-131): DALIGNER's local alignment around the anchor's diagonal
// (ocda_go, gapped_align/oc_daligner.c:36 -> Local_Alignment, gapped_align/align.c:1754, forward_wave :382, reverse_wave :1043),
// then a global alignment of exactly that range with its path (edlib_go, edlib/edlib_wrapper.c:118).  Host code, as in the
// reference: the pair runs for the few candidates whose extension fell short, not on the hot path.
//
// Local_Alignment restated: a furthest-reaching-point wave over diagonals k = a - b, V[k] = a + b of the furthest point on k with
// `dif` differences, started on the anti-diagonal a + b = anti; every point carries the match / mismatch history of its last 60
// columns (T, M) - a wave point only extends the "good" tip while at least ave_path of them match, and the path is trimmed back to
// the last tip whose two latest 15-column windows score non-negative (table / score) - and waves stop TRIM_MLAG past the last such
// tip or when a sequence end is reached.  The reference also threads trace points ("pebbles") through the wave; its callers here
// only read the end points and the difference count, and of the pebble chains only the ROOT matters to them (the diagonal the
// winning path started on: the reverse wave starts there), so every point carries that diagonal (org) instead.
#pragma once
#include <cstdint>
#include <climits>
#include <cstddef>
#include <vector>

namespace rescue {

constexpr int kTrimLen = 15, kPathLen = 60;
constexpr uint64_t kPathTop = 1ULL << kPathLen, kPathInt = kPathTop - 1;
constexpr int kTrimMask = (1 << kTrimLen) - 1, kTrimMlag = 200, kWaveLag = 30;

struct DalignSpec {          // New_Align_Spec (align.c:257-301)
    int ave_path = 0;
    std::vector<int16_t> table, score;
};

inline DalignSpec make_spec(double ave_corr, const float freq[4])
{
    static const double bias_factor[10] = {.690, .690, .690, .690, .780, .850, .900, .933, .966, 1.000};
    DalignSpec s;
    double match = (double)(freq[0] + freq[3]);
    if (match > .5) match = 1. - match;
    int bias = (int)((match + .025) * 20. - 1.);
    if (match < .2) bias = 3;
    s.ave_path = (int)(kPathLen * (1. - bias_factor[bias] * (1. - ave_corr)));
    const int mscore = (int)(1000 * bias_factor[bias] * (1. - ave_corr)), dscore = 1000 - mscore;
    s.table.assign(kTrimMask + 1, 0); s.score.assign(kTrimMask + 1, 0);
    // a 15-column window read from its most significant bit (1 = match): score = its total, table = total - the best prefix total
    for (int x = 0; x <= kTrimMask; ++x) {
        int sc = 0, mx = 0;
        for (int bit = kTrimLen - 1; bit >= 0; --bit) { if (sc > mx) mx = sc; sc += ((x >> bit) & 1) ? mscore : -dscore; }
        s.table[x] = (int16_t)(sc - mx); s.score[x] = (int16_t)sc;
    }
    return s;
}

struct DalignResult { int abpos = 0, aepos = 0, bbpos = 0, bepos = 0, diffs = 0; };

namespace detail {

struct Wave {               // per-diagonal state, addressed by k in [kmin, kmax]
    int kmin = 0;
    std::vector<int> V, M, O;
    std::vector<uint64_t> T;
    void reset(int lo, int hi) { kmin = lo; const size_t n = (size_t)(hi - lo + 1); V.assign(n, 0); M.assign(n, 0); O.assign(n, 0); T.assign(n, 0); }
    int& v(int k) { return V[(size_t)(k - kmin)]; }
    int& m(int k) { return M[(size_t)(k - kmin)]; }
    int& o(int k) { return O[(size_t)(k - kmin)]; }
    uint64_t& t(int k) { return T[(size_t)(k - kmin)]; }
};

struct Tip { int a, y, d, org; };      // a = x + y of the point, y its b coordinate, d differences, org the diagonal its path started on

// one direction of Local_Alignment.  DIR = +1: forward_wave (align.c:382), towards the sequence ends; DIR = -1: reverse_wave
// (align.c:1043), towards their starts, comparing the characters BEFORE a position.  aseq / bseq: base codes with the sentinel 4
// at [-1] and [len].  Returns the trimmed tip (or the best point that ran into a sequence end).
template <int DIR>
inline Tip wave(const char* aseq_in, const char* bseq_in, const DalignSpec& S, Wave& W, int low, int hgh, int mida, int minp, int maxp)
{
    const char* aseq = DIR > 0 ? aseq_in : aseq_in - 1;
    const char* bseq = DIR > 0 ? bseq_in : bseq_in - 1;
    const int kNone = DIR > 0 ? -1 : INT32_MAX;             // "no point yet" on a fresh diagonal
    auto better = [](int x, int y) { return DIR > 0 ? x > y : x < y; };      // x is further along than y
    int dif = 0, more = 1;
    int aclip = DIR > 0 ? INT32_MAX : -INT32_MAX, bclip = DIR > 0 ? -INT32_MAX : INT32_MAX;
    int besta = mida, lasta = mida, besty = (mida - hgh) >> 1;
    const int first = DIR > 0 ? hgh : low;               // the reference's pebble 0: the first diagonal wave 0 visits
    Tip trim{mida, besty, 0, first}, mor{mida, besty, 0, first};
    int morem = -1;
    // slide along diagonal k from (y + k, y) while the characters agree; stops at the sentinels
    auto slide = [&](int k, int& y, int& m, uint64_t& b, bool history) {
        const char* a = aseq + k;
        for (;;) {
            const int c = bseq[y];
            if (c == 4) { more = 0; if (DIR > 0 ? bclip < k : bclip > k) bclip = k; break; }
            const int d = a[y];
            if (c != d) { if (d == 4) { more = 0; aclip = k; } break; }
            y += DIR;
            if (history) { if ((b & kPathTop) == 0) m += 1; b = (b << 1) | 1; }
        }
    };
    // a diagonal that ran into a sequence end leaves the wave; the best such point is remembered
    auto clip = [&]() {
        if (bseq[besty] != 4 && aseq[besta - besty] != 4) more = 1;
        const bool a_hit = DIR > 0 ? hgh >= aclip : low <= aclip, b_hit = DIR > 0 ? low <= bclip : hgh >= bclip;
        if (a_hit) {
            if (DIR > 0) hgh = aclip - 1; else low = aclip + 1;
            if (morem <= W.m(aclip)) { morem = W.m(aclip); mor.a = W.v(aclip); mor.y = (mor.a - aclip) / 2; mor.d = dif; mor.org = W.o(aclip); }
        }
        if (b_hit) {
            if (DIR > 0) low = bclip + 1; else hgh = bclip - 1;
            if (morem <= W.m(bclip)) { morem = W.m(bclip); mor.a = W.v(bclip); mor.y = (mor.a - bclip) / 2; mor.d = dif; mor.org = W.o(bclip); }
        }
        aclip = DIR > 0 ? INT32_MAX : -INT32_MAX; bclip = DIR > 0 ? -INT32_MAX : INT32_MAX;
    };

    // wave 0: the snakes from the anti-diagonal `mida`, visited the way the reference does (downwards when going forward)
    for (int i = 0; i <= hgh - low; ++i) {
        const int k = DIR > 0 ? hgh - i : low + i;
        int y = (mida - k) >> 1, m = 0; uint64_t b = 0;
        slide(k, y, m, b, false);
        const int c = (y << 1) + k;
        if (better(c, besta)) { besta = trim.a = lasta = c; besty = trim.y = y; trim.org = k; }
        W.v(k) = c; W.t(k) = kPathInt; W.m(k) = kPathLen; W.o(k) = k;
    }
    if (more == 0) clip();

    while (more && (DIR > 0 ? lasta >= besta - kTrimMlag : lasta <= besta + kTrimMlag)) {
        low -= 1; hgh += 1;
        // the two fresh diagonals (unless outside [minp, maxp]); the neighbour beyond the wave counts as "no point"
        int edge = kNone;        // the not-yet-updated value of the diagonal the sweep starts next to
        if (DIR > 0) {
            if (low >= minp) W.v(low) = kNone; else low += 1;
            if (hgh <= maxp) { W.v(hgh) = kNone; edge = kNone; } else edge = W.v(--hgh);
        } else {
            if (low >= minp) { W.v(low) = kNone; edge = kNone; } else edge = W.v(++low);
            if (hgh <= maxp) W.v(hgh) = kNone; else hgh -= 1;
        }
        dif += 1;
        W.v(hgh + 1) = kNone; W.v(low - 1) = kNone;
        // sweep the diagonals (downwards when going forward); `prev*` = the diagonal just left, BEFORE this wave updated it
        int ac = kNone, nxt = edge;              // ac: current diagonal's old value, nxt: the next diagonal's old value
        int prev_v = kNone;
        uint64_t prev_t = kPathInt; int prev_m = kPathLen, prev_o = -1;
        for (int i = 0; i <= hgh - low; ++i) {
            const int k = DIR > 0 ? hgh - i : low + i;
            const int d = k - DIR;                      // the diagonal the sweep reaches next (still holding the previous wave)
            prev_v = ac; ac = nxt; nxt = W.v(d);
            // the furthest of: one step from the diagonal just left (prev), one step from the next one (nxt), two along this one
            int c, m, o; uint64_t b;
            const bool use_prev = better(nxt, ac) ? better(prev_v, nxt) : better(prev_v, ac);
            if (use_prev) { c = prev_v + DIR; m = prev_m; b = prev_t; o = prev_o; }
            else if (better(nxt, ac)) { c = nxt + DIR; m = W.m(d); b = W.t(d); o = W.o(d); }
            else { c = ac + 2 * DIR; m = W.m(k); b = W.t(k); o = W.o(k); }
            if (b & kPathTop) m -= 1;
            b <<= 1;
            int y = (c - k) >> 1;
            slide(k, y, m, b, true);
            c = (y << 1) + k;
            if (better(c, besta)) {
                besta = c; besty = y;
                if (m >= S.ave_path) {
                    lasta = c;
                    if (S.table[b & kTrimMask] >= 0 && S.table[(b >> kTrimLen) & kTrimMask] + S.score[b & kTrimMask] >= 0) { trim.a = c; trim.y = y; trim.d = dif; trim.org = o; }
                }
            }
            prev_t = W.t(k); prev_m = W.m(k); prev_o = W.o(k);
            W.v(k) = c; W.t(k) = b; W.m(k) = m; W.o(k) = o;
        }
        if (more == 0) clip();
        // points more than WAVE_LAG behind the best one leave the wave
        const int lim = besta - DIR * kWaveLag;
        while (hgh >= low) {
            if (better(lim, W.v(hgh))) hgh -= 1;
            else { while (better(lim, W.v(low))) low += 1; break; }
        }
    }
    if (morem >= 0) return mor;
    return trim;
}

}  // namespace detail

// Local_Alignment(align, work, spec, low, hgh, anti, -1, -1) for two different sequences, no complement flags (what ocda_go calls):
// aseq / bseq as described above (sentinels in place).
inline DalignResult local_alignment(const char* aseq, int alen, const char* bseq, int blen, const DalignSpec& S, int low, int hgh, int anti)
{
    detail::Wave W;
    W.reset(-blen - 4, alen + 4);
    const int minp = -INT32_MAX, maxp = INT32_MAX;
    const detail::Tip f = detail::wave<+1>(aseq, bseq, S, W, low, hgh, anti, minp, maxp);
    DalignResult r;
    r.aepos = f.a - f.y; r.bepos = f.y; r.diffs = f.d;
    const int org = f.org;
    const detail::Tip b = detail::wave<-1>(aseq, bseq, S, W, org, org, anti, minp, maxp);
    r.abpos = b.a - b.y; r.bbpos = b.y; r.diffs += b.d;
    return r;
}

// ocda_go (gapped_align/oc_daligner.c:36-79): the local alignment through the anchor (query_start, target_start); base codes 0..3
struct Dalign {
    DalignSpec spec;
    DalignResult r;
    double ident_perc = 0.0;
    std::vector<char> a, b;
    explicit Dalign(double error) { const float f[4] = {.25f, .25f, .25f, .25f}; spec = make_spec(1.0 - error, f); }
    bool go(const char* query, int query_start, int query_size, const char* target, int target_start, int target_size, int min_align_size)
    {
        a.assign((size_t)query_size + 2, 4); b.assign((size_t)target_size + 2, 4);
        for (int i = 0; i < query_size; ++i) a[(size_t)i + 1] = query[i];
        for (int i = 0; i < target_size; ++i) b[(size_t)i + 1] = target[i];
        r = local_alignment(a.data() + 1, query_size, b.data() + 1, target_size, spec, query_start - target_start, query_start - target_start,
                            query_start + target_start);
        const int asize = r.aepos - r.abpos, bsize = r.bepos - r.bbpos;
        if (!(asize >= min_align_size && bsize >= min_align_size)) return false;
        ident_perc = 100.0 - 200.0 * r.diffs / (asize + bsize);
        return true;
    }
};

}  // namespace rescue
