// host_fmt.h - the record writers of oc2pmov / oc2pm: the reference's printf formats (m4_record.h:72-124, gapped_candidate.h:26-42)
// produced without printf.  228 k M4 lines through fprintf cost 70 ms, most of it in "%.2f"; here integers are written digit by
// digit and the identity is rounded exactly as printf does: with x = M * 2^-s (M < 2^53), x * 100 = 100 M / 2^s is an exact
// integer ratio, so quotient, remainder and round-half-even on it reproduce glibc's correctly rounded "%.2f".
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <string>
#include <thread>
#include <vector>
#include "../../include/necat_hip.h"

namespace necat_host {

inline char* put_u64(char* p, uint64_t v)
{
    char tmp[24]; int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
inline char* put_i64(char* p, int64_t v)
{
    if (v < 0) { *p++ = '-'; return put_u64(p, (uint64_t)0 - (uint64_t)v); }
    return put_u64(p, (uint64_t)v);
}
// "%.2f"
inline char* put_f2(char* p, double x)
{
    if (!(x == x) || x < 0 || x >= 4503599627370496.0) { p += sprintf(p, "%.2f", x); return p; }     // NaN / negative / >= 2^52: not produced by the aligner
    int e; const double fr = frexp(x, &e);                     // x = fr * 2^e, fr in [0.5, 1)
    const uint64_t M = (uint64_t)ldexp(fr, 53);                // 53-bit integer mantissa (exact)
    const int s = 53 - e;                                      // x = M * 2^-s, s >= 1 here
    uint64_t r = 0;
    if (s <= 62) {
        const uint64_t N = 100 * M;                            // < 2^60
        r = N >> s;
        const uint64_t rem = N & ((1ULL << s) - 1), half = 1ULL << (s - 1);
        if (rem > half || (rem == half && (r & 1ULL))) ++r;    // round half to even
    }                                                          // else x * 100 < 2^-2: "0.00"
    p = put_u64(p, r / 100);
    *p++ = '.'; *p++ = (char)('0' + (r / 10) % 10); *p++ = (char)('0' + r % 10);
    return p;
}
inline char* put_str(char* p, const char* s) { const size_t n = strlen(s); memcpy(p, s, n); return p + n; }

// DUMP_ASM_M4 (m4_record.h:72-97); qname / sname != nullptr: DUMP_ASM_M4_HDR_ID (m4_record.h:99-124)
inline char* put_m4(char* p, const necat_m4& m, const char* qname, const char* sname)
{
    if (qname) { p = put_str(p, qname); *p++ = '\t'; p = put_str(p, sname); } else { p = put_i64(p, m.qid); *p++ = '\t'; p = put_i64(p, m.sid); }
    *p++ = '\t'; p = put_f2(p, m.ident_perc);
    *p++ = '\t'; p = put_i64(p, m.vscore); *p++ = '\t'; p = put_i64(p, m.qdir);
    *p++ = '\t'; p = put_u64(p, m.qoff); *p++ = '\t'; p = put_u64(p, m.qend); *p++ = '\t'; p = put_u64(p, m.qsize);
    *p++ = '\t'; p = put_i64(p, m.sdir);
    *p++ = '\t'; p = put_u64(p, m.soff); *p++ = '\t'; p = put_u64(p, m.send); *p++ = '\t'; p = put_u64(p, m.ssize);
    *p++ = '\n';
    return p;
}
// DUMP_GAPPED_CANDIDATE (gapped_candidate.h:26-42)
inline char* put_candidate(char* p, const necat_candidate& c)
{
    p = put_i64(p, c.qid); *p++ = '\t'; p = put_i64(p, c.sid); *p++ = '\t'; p = put_i64(p, c.score); *p++ = '\t'; p = put_i64(p, c.qdir);
    *p++ = '\t'; p = put_u64(p, c.qbeg); *p++ = '\t'; p = put_u64(p, c.qend); *p++ = '\t'; p = put_u64(p, c.qoff); *p++ = '\t'; p = put_u64(p, c.qsize);
    *p++ = '\t'; p = put_i64(p, c.sdir);
    *p++ = '\t'; p = put_u64(p, c.sbeg); *p++ = '\t'; p = put_u64(p, c.send); *p++ = '\t'; p = put_u64(p, c.soff); *p++ = '\t'; p = put_u64(p, c.ssize);
    *p++ = '\n';
    return p;
}

// n records formatted by up to `threads` host threads (record order kept), written with one fwrite per thread's share.
// put(p, i) appends record i at p and returns the new end; max_len bounds one record's text.
template <class Put>
inline bool write_records(FILE* out, uint64_t n, size_t max_len, int threads, Put put)
{
    if (n == 0) return true;
    int T = threads < 1 ? 1 : (threads > 32 ? 32 : threads);
    if (n < 4096) T = 1;
    std::vector<std::vector<char>> buf(T);
    auto work = [&](int t) {
        const uint64_t lo = n * t / T, hi = n * (t + 1) / T;
        buf[t].resize((size_t)(hi - lo) * max_len + 16);
        char* p = buf[t].data();
        for (uint64_t i = lo; i < hi; ++i) p = put(p, i);
        buf[t].resize((size_t)(p - buf[t].data()));
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    bool ok = true;
    for (int t = 0; t < T; ++t) if (!buf[t].empty()) ok = fwrite(buf[t].data(), 1, buf[t].size(), out) == buf[t].size() && ok;
    return ok;
}

}  // namespace necat_host
