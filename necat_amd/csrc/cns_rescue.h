// cns_rescue.h - the second half of cns_extension (consensus/consensus_aux.c:124-215) for oc2cns -r 1 (rescue_long_indels):
// a candidate whose block-wise extension failed, or stopped more than 200 bp short of the candidate's chained range in the
// query, is aligned again around its anchor with DALIGNER's local alignment and, if that gives an overlap, globally over exactly
// that range with edlib's path (rescue.h); the block-wise result is kept when the pair gives nothing.
//
// Host code as in the reference, run after a device pass on the candidates that need it (stage_cns.inl; tests/host_core/
// check_cns.cpp plugs it behind the oracle's aligner).  No HIP in this header.
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/necat_hip.h"
#include "rescue.h"

namespace necat {
namespace cns {

// consensus_aux.c:152-159: is the block-wise result (ok / coordinates) final?
inline bool extension_short(const necat_candidate& c, const necat_alignment& a)
{
    if (!a.ok) return true;
    const int qbeg = (int)c.qbeg, qend = (int)c.qend;
    const int lhang = a.qoff > qbeg ? a.qoff - qbeg : 0;
    const int rhang = a.qend < qend ? qend - a.qend : 0;
    return lhang + rhang > 200;
}

struct Rescuer {
    rescue::Dalign dal;
    rescue::EdlibGo edl;
    std::vector<uint8_t> cols;      // the rescued alignment's columns, one code per column as in necat_onc_align_batch
    Rescuer(const rescue::DalignSpec& spec, double error) : dal(spec), edl(error) {}

    // qstrand: the query read on the candidate's strand, tseq: the template, base codes 0..3.  True: *a and cols hold the pair's
    // alignment (consensus_aux.c:170-199); false: the block-wise result stands.
    bool go(const necat_candidate& c, const uint8_t* qstrand, const uint8_t* tseq, int min_align_size, necat_alignment* a)
    {
        if (!dal.go((const char*)qstrand, (int)c.qoff, (int)c.qsize, (const char*)tseq, (int)c.soff, (int)c.ssize, min_align_size)) return false;
        if (!edl.go((const char*)qstrand, dal.r.abpos, dal.r.aepos, (const char*)tseq, dal.r.bbpos, dal.r.bepos, dal.r.diffs, min_align_size)) return false;
        const size_t n = edl.query_align.size();
        cols.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const char q = edl.query_align[i], t = edl.target_align[i];
            cols[i] = q == '-' ? 2 : (t == '-' ? 1 : (q == t ? 0 : 3));
        }
        a->ok = 1; a->qoff = edl.qoff; a->qend = edl.qend; a->toff = edl.toff; a->tend = edl.tend;
        a->align_size = (int32_t)n; a->ident_perc = edl.ident_perc;
        return true;
    }
};

}  // namespace cns
}  // namespace necat
