/* rescue_ref_shim.c - TEST INFRASTRUCTURE.  Flat entry points over the REFERENCE's own rescue pair, as linked into
 * _ref/libnecat_cns_ref.so from the sources under /root/reference (nothing of them is copied here):
 *   ocda_go  (gapped_align/oc_daligner.c:36, DALIGNER's Local_Alignment, gapped_align/align.c:1754)
 *   edlib_go (edlib/edlib_wrapper.c:118, edlibAlign in NW mode with its path, edlib/edlib.cpp)
 * Sequences are base codes 0..3, as the reference's callers hand them over (consensus_aux.c:170-195, rm_worker.c:104-131). */
#include "gapped_align/oc_daligner.h"
#include "edlib/edlib_wrapper.h"
#include <string.h>

/* out: abpos, aepos, bbpos, bepos, diffs; *ident = ident_perc.  Returns ocda_go's BOOL. */
int ref_ocda_go(const char* query, int query_start, int query_size, const char* target, int target_start, int target_size,
                double error, int min_align_size, int* out, double* ident)
{
    OcDalignData* d = new_OcDalignData(error);
    const int r = ocda_go(query, query_start, query_size, target, target_start, target_size, d, min_align_size);
    out[0] = ocda_query_start(*d); out[1] = ocda_query_end(*d); out[2] = ocda_target_start(*d); out[3] = ocda_target_end(*d);
    out[4] = ocda_distance(*d);
    *ident = ocda_ident_perc(*d);
    free_OcDalignData(d);
    return r;
}

/* out: qoff, qend, toff, tend, dist, alignment length; qaln / taln (may be NULL): the trimmed alignment strings (capacity cap).
 * Returns edlib_go's int. */
int ref_edlib_go(const char* query, int query_from, int query_to, const char* target, int target_from, int target_to,
                 double error, int tolerance, int min_align_size, int* out, double* ident, char* qaln, char* taln, int cap)
{
    FullEdlibAlignData* d = new_FullEdlibAlignData(error);
    const int r = edlib_go(query, query_from, query_to, target, target_from, target_to, d, tolerance, min_align_size, TRUE, 4);
    if (r) {
        out[0] = d->qoff; out[1] = d->qend; out[2] = d->toff; out[3] = d->tend; out[4] = d->dist;
        const int n = (int)kstr_size(d->query_align);
        out[5] = n;
        *ident = d->ident_perc;
        if (qaln && taln && n < cap) { memcpy(qaln, kstr_str(d->query_align), n); qaln[n] = 0; memcpy(taln, kstr_str(d->target_align), n); taln[n] = 0; }
    }
    free_FullEdlibAlignData(d);
    return r;
}
