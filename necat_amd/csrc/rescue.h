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
#include <algorithm>
#include <string>
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
inline DalignSpec spec_for_error(double error) { const float f[4] = {.25f, .25f, .25f, .25f}; return make_spec(1.0 - error, f); }   // oc_daligner.c:7-17

struct Dalign {
    DalignSpec own;
    const DalignSpec& spec;
    DalignResult r;
    double ident_perc = 0.0;
    std::vector<char> a, b;
    explicit Dalign(double error) : own(spec_for_error(error)), spec(own) {}
    explicit Dalign(const DalignSpec& shared) : spec(shared) {}      // many workers, one pair of 32 K-entry tables
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

// ---------------------------------------------------------------------------------------------------------------------------
// edlib_go (edlib/edlib_wrapper.c:111-242): the global alignment of query[query_from, query_to) with target[target_from,
// target_to), accepted when its distance is at most `tolerance` and at most error * (target_size - 1), as the path edlib's NW mode
// returns for it (edlib/edlib.cpp: edlibAlign :149, obtainAlignment :1106, obtainAlignmentHirschberg :1177,
// obtainAlignmentTraceback :887), trimmed at both ends to the first run of kMatchSize matches.
//
// Among the optimal alignments edlib returns one particular path, and the callers read the alignment strings, so the choice is
// restated with it: a problem whose traceback store (20 bytes per 64-row block and column + 8 per column) stays under 1 MB is
// walked back from its last cell preferring "up" (a query base against a gap), then "left" (a target base against a gap), then
// the diagonal; a larger one is split at the middle column of the target, at the FIRST query row (ascending; then the row above
// the first, then the last) where the prefix cost to the left column plus the suffix cost from the next row and column add up to
// the optimum, and both parts are solved the same way with those two costs as their optima.  Every cell either rule can select lies
// on an optimal path, so any computation that is exact on the cells of paths within the optimum - edlib's banded blocks, the
// banded columns below - makes the same choices (the tests pin this against the reference's build).
namespace detail {

enum : uint8_t { kOpMatch = 0, kOpInsert = 1, kOpDelete = 2, kOpMismatch = 3 };   // EDLIB_EDOP_*

constexpr int kFar = INT32_MAX / 4;     // "outside the band"

// Myers' bit-vector columns, 64 query rows per word, inside Ukkonen's band for an optimum of at most k: a cell (r, c) of a path
// that costs at most k in total has |r - c| + |(m - n) - (r - c)| <= k.  Whatever lies outside enters as an upper bound (a new
// block below starts one more per row than the block above it, the row above the band one more per column), so every cell whose
// true value can matter - those on a path within k - comes out exact, the others at least as large as they are.
struct BitColumns {
    int m = 0, nb = 0, dmin = 0, dmax = 0, fb = 0, lb = -1;
    std::vector<uint64_t> eq, P, M;
    std::vector<int> bot;            // value at the last row of each block in the current column
    std::vector<uint64_t> sv, sh;    // traceback store: per column and block, the vertical (after the column) and horizontal +1 flags
    void set_query(const uint8_t* q, int m_)
    {
        m = m_; nb = (m + 63) / 64;
        eq.assign((size_t)4 * nb, 0);
        for (int r = 0; r < m; ++r) eq[(size_t)(q[r] & 3) * nb + (r >> 6)] |= 1ULL << (r & 63);
    }
    void set_band(int n_total, int k) { const int delta = m - n_total; dmin = -((k - delta) / 2); dmax = (delta + k) / 2; }
    // columns 0 .. n_stop - 1 of the target
    template <bool STORE>
    void run(const uint8_t* t, int n_stop)
    {
        P.resize((size_t)nb); M.resize((size_t)nb); bot.resize((size_t)nb);
        if (STORE) { sv.resize((size_t)n_stop * nb); sh.resize((size_t)n_stop * nb); }
        lb = -1; fb = 0;
        for (int c = 0; c < n_stop; ++c) {
            const int nfb = std::max(0, c + dmin) >> 6, nlb = std::min(m - 1, c + dmax) >> 6;
            for (int b = lb + 1; b <= nlb; ++b) { P[b] = ~0ULL; M[b] = 0; bot[b] = (b ? bot[b - 1] : c) + 64; }
            fb = nfb; lb = nlb;
            const uint64_t* e = &eq[(size_t)(t[c] & 3) * nb];
            int hin = 1;
            for (int b = fb; b <= lb; ++b) {
                uint64_t pv = P[b], mv = M[b], x = e[b];
                const uint64_t hneg = hin < 0 ? 1ULL : 0ULL;
                const uint64_t xv = x | mv;
                x |= hneg;
                const uint64_t xh = (((x & pv) + pv) ^ pv) | x;
                uint64_t ph = mv | ~(xh | pv), mh = pv & xh;
                const int hout = (int)(ph >> 63) - (int)(mh >> 63);
                if (STORE) sh[(size_t)c * nb + b] = ph;
                ph = (ph << 1) | (hin > 0 ? 1ULL : 0ULL); mh = (mh << 1) | hneg;
                P[b] = mh | ~(xv | ph); M[b] = ph & xv;
                if (STORE) sv[(size_t)c * nb + b] = P[b];
                bot[b] += hout;
                hin = hout;
            }
        }
    }
    // D(r, n_stop - 1) for r in [0, m) after run(): the cost of query[0..r] against target[0..n_stop-1]; kFar outside the band
    void column(std::vector<int>& col) const
    {
        col.assign((size_t)m, kFar);
        for (int b = fb; b <= lb; ++b) {
            int s = bot[b];
            for (int r = 64 * b + 63; r >= 64 * b; --r) {
                if (r < m) col[(size_t)r] = s;
                s -= (int)((P[b] >> (r & 63)) & 1) - (int)((M[b] >> (r & 63)) & 1);
            }
        }
    }
};

struct NwPath {
    std::vector<uint8_t> ops;
    std::vector<int> L, R;
    std::vector<uint8_t> rq, rt;
    BitColumns bc;

    // obtainAlignmentTraceback's walk.  Its three tests read, for a cell on an optimal path: is the cell above one less (the
    // vertical +1 flag), else is the cell to the left one less (the horizontal +1 flag), else the diagonal - a match exactly when
    // the two bases are equal, since neither neighbour being one less leaves "diagonal + mismatch" or "diagonal, equal bases".
    void leaf(const uint8_t* q, int m, const uint8_t* t, int n, int best)
    {
        bc.set_query(q, m);
        bc.set_band(n, best);
        bc.run<true>(t, n);
        const size_t at0 = ops.size();
        const int nb = bc.nb;
        int r = m - 1, c = n - 1;
        while (r >= 0 && c >= 0) {
            const size_t w = (size_t)c * nb + (r >> 6);
            if ((bc.sv[w] >> (r & 63)) & 1) { ops.push_back(kOpInsert); --r; }
            else if ((bc.sh[w] >> (r & 63)) & 1) { ops.push_back(kOpDelete); --c; }
            else { ops.push_back(q[r] == t[c] ? kOpMatch : kOpMismatch); --r; --c; }
        }
        for (; r >= 0; --r) ops.push_back(kOpInsert);
        for (; c >= 0; --c) ops.push_back(kOpDelete);
        std::reverse(ops.begin() + (ptrdiff_t)at0, ops.end());
    }

    // obtainAlignment: appends the path of q[0, m) against t[0, n), whose optimum is `best`
    bool solve(const uint8_t* q, int m, const uint8_t* t, int n, int best)
    {
        if (m == 0 || n == 0) { ops.insert(ops.end(), (size_t)(m + n), m == 0 ? kOpDelete : kOpInsert); return true; }
        const long long blocks = (m + 63) / 64;
        if (20LL * blocks * n + 8LL * n < 1024 * 1024) { leaf(q, m, t, n, best); return true; }
        const int lw = n / 2, rw = n - lw;
        bc.set_query(q, m);
        bc.set_band(n, best);
        bc.run<false>(t, lw);
        bc.column(L);
        rq.assign(q, q + m); std::reverse(rq.begin(), rq.end());
        rt.assign(t + lw, t + n); std::reverse(rt.begin(), rt.end());
        bc.set_query(rq.data(), m);
        bc.set_band(n, best);
        bc.run<false>(rt.data(), rw);
        bc.column(R);                                // R[i]: query[m-1-i, m) against the right half
        int row = -2, ls = 0, rs = 0;
        for (int i = 0; i + 1 < m; ++i)
            if (L[(size_t)i] + R[(size_t)(m - 2 - i)] == best) { row = i; ls = L[(size_t)i]; rs = R[(size_t)(m - 2 - i)]; break; }
        if (row == -2 && lw + R[(size_t)m - 1] == best) { row = -1; ls = lw; rs = R[(size_t)m - 1]; }
        if (row == -2 && L[(size_t)m - 1] + rw == best) { row = m - 1; ls = L[(size_t)m - 1]; rs = rw; }
        if (row == -2) return false;
        const int uh = row + 1;
        return solve(q, uh, t, lw, ls) && solve(q + uh, m - uh, t + lw, rw, rs);
    }
};

}  // namespace detail

struct EdlibGo {
    double error;
    int qoff = 0, qend = 0, toff = 0, tend = 0, dist = 0;
    double ident_perc = 0.0;
    std::string query_align, target_align;       // gapped, 'A' 'C' 'G' 'T' '-'
    detail::NwPath path;
    std::vector<int> col;
    std::string qa, ta;
    explicit EdlibGo(double error_) : error(error_) {}

    // find_path = TRUE, as consensus_aux.c:179 and rm_worker.c:115 call it
    bool go(const char* query, int query_from, int query_to, const char* target, int target_from, int target_to, int tolerance,
            int min_align_size, int match_size = 4)
    {
        const int m = query_to - query_from, n = target_to - target_from;
        if (m <= 0 || n <= 0 || tolerance < 0) return false;
        const uint8_t* q = (const uint8_t*)query + query_from;
        const uint8_t* t = (const uint8_t*)target + target_from;
        if (tolerance < (m > n ? m - n : n - m)) return false;
        path.bc.set_query(q, m);
        path.bc.set_band(n, std::min(tolerance, std::max(m, n)));
        path.bc.run<false>(t, n);
        path.bc.column(col);
        const int best = col[(size_t)m - 1];
        if (best > tolerance) return false;
        const int align_len = n - 1;
        if (align_len < min_align_size) return false;
        if ((double)best / (double)align_len > error) return false;
        path.ops.clear();
        if (!path.solve(q, m, t, n, best)) return false;
        const int len = (int)path.ops.size();
        qa.resize((size_t)len); ta.resize((size_t)len);
        for (int a = 0, x = 0, y = 0; a < len; ++a) {
            const uint8_t op = path.ops[(size_t)a];
            if (op == detail::kOpMatch || op == detail::kOpMismatch) { qa[(size_t)a] = "ACGT"[q[x++] & 3]; ta[(size_t)a] = "ACGT"[t[y++] & 3]; }
            else if (op == detail::kOpInsert) { qa[(size_t)a] = "ACGT"[q[x++] & 3]; ta[(size_t)a] = '-'; }
            else { qa[(size_t)a] = '-'; ta[(size_t)a] = "ACGT"[t[y++] & 3]; }
        }
        // both ends are cut back to their first run of match_size matches
        int from = 0, pq = 0, pt = 0, run = 0;
        while (from < len) {
            run = qa[(size_t)from] == ta[(size_t)from] ? run + 1 : 0;
            if (qa[(size_t)from] != '-') ++pq;
            if (ta[(size_t)from] != '-') ++pt;
            ++from;
            if (run == match_size) break;
        }
        if (run != match_size) return false;
        from -= match_size; pq -= match_size; pt -= match_size;
        int to = len, tq = 0, tt = 0;
        run = 0;
        while (to) {
            run = qa[(size_t)to - 1] == ta[(size_t)to - 1] ? run + 1 : 0;
            if (qa[(size_t)to - 1] != '-') ++tq;
            if (ta[(size_t)to - 1] != '-') ++tt;
            --to;
            if (run == match_size) break;
        }
        to += match_size; tq -= match_size; tt -= match_size;
        query_align.assign(qa, (size_t)from, (size_t)(to - from));
        target_align.assign(ta, (size_t)from, (size_t)(to - from));
        qoff = query_from + pq; qend = query_from + m - tq;
        toff = target_from + pt; tend = target_from + n - tt;
        const int asz = to - from;
        int same = 0;
        for (int i = 0; i < asz; ++i) same += query_align[(size_t)i] == target_align[(size_t)i];
        if (asz == 0) { ident_perc = 0.0; }
        else { dist = asz - same; ident_perc = 100.0 * same / asz; }
        return true;
    }
};

}  // namespace rescue
