// flat entry points over necat_amd/csrc/rescue.h for the CPU parity tests (tests/test_rescue.py): same signatures as
// oracle/rescue_ref_shim.c gives the reference's functions
#include "../../necat_amd/csrc/rescue.h"
#include <cstring>
#include <string>

extern "C" int mine_ocda_go(const char* query, int query_start, int query_size, const char* target, int target_start, int target_size,
                            double error, int min_align_size, int* out, double* ident)
{
    rescue::Dalign d(error);
    const bool ok = d.go(query, query_start, query_size, target, target_start, target_size, min_align_size);
    out[0] = d.r.abpos; out[1] = d.r.aepos; out[2] = d.r.bbpos; out[3] = d.r.bepos; out[4] = d.r.diffs;
    *ident = d.ident_perc;
    return ok ? 1 : 0;
}

extern "C" int mine_edlib_go(const char* query, int query_from, int query_to, const char* target, int target_from, int target_to,
                             double error, int tolerance, int min_align_size, int* out, double* ident, char* qaln, char* taln, int cap)
{
    rescue::EdlibGo e(error);
    if (!e.go(query, query_from, query_to, target, target_from, target_to, tolerance, min_align_size)) return 0;
    out[0] = e.qoff; out[1] = e.qend; out[2] = e.toff; out[3] = e.tend; out[4] = e.dist;
    const int n = (int)e.query_align.size();
    out[5] = n;
    *ident = e.ident_perc;
    if (qaln && taln && n < cap) { memcpy(qaln, e.query_align.data(), (size_t)n); qaln[n] = 0; memcpy(taln, e.target_align.data(), (size_t)n); taln[n] = 0; }
    return 1;
}
