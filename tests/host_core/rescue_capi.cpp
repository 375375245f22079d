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
