/*
 * cns_oracle.c - CPU restatement of the EXTENSION LOOP of NECAT's consensus stage (oc2cns), SURVEY §8f.1:
 * which candidates of a template get aligned, in what order, and which alignments are handed to the
 * consensus (add_one_align) with what weight.  The consensus itself (tasc/cbcns.c) is not restated.
 *
 * TEST INFRASTRUCTURE ONLY - same rules as necat_oracle.c (only tests/, smoke() and bench.py's cpu_baseline
 * leg may use it, as the checker).  Pinned against the reference's own consensus_one_partition run through
 * oracle/cns_ref_harness.c (tests/test_oracle_golden.py::test_cns_loop_*; tests/golden/cns_c/ref_*.txt).
 *
 * Reference paths are relative to /root/reference/src/.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "necat_oracle.h"

/* ---- candidate records between oc2pmov -j 0 and oc2cns ---- */

/* common/gapped_candidate.c:31-52 unpack_candidate */
void ora_unpack_candidate(const uint32_t item[7], ora_candidate* c)
{
    memset(c, 0, sizeof *c);
    c->sdir = (int)(item[0] >> 31) & 1;
    c->qdir = (int)(item[0] >> 30) & 1;
    c->score = (int)(item[0] & ((1u << 29) - 1));
    c->sid = (int32_t)item[1]; c->sbeg = item[2]; c->send = item[3];
    c->qid = (int32_t)item[4]; c->qbeg = item[5]; c->qend = item[6];
    if (item[0] & (1u << 29)) { c->qoff = c->qbeg; c->soff = c->sbeg; }
    else { c->qoff = c->qend; c->soff = c->send; }
}

/* common/gapped_candidate.c:54-69 change_pcan_roles: the subject becomes the query and vice versa */
void ora_change_pcan_roles(const uint32_t src[7], uint32_t dst[7])
{
    uint32_t flags = 0;
    if (src[0] >> 31) flags |= 1u << 30;
    if ((src[0] >> 30) & 1) flags |= 1u << 31;
    flags |= src[0] & (1u << 29);
    flags |= src[0] & ((1u << 29) - 1);
    dst[0] = flags;
    dst[1] = src[4]; dst[2] = src[5]; dst[3] = src[6];
    dst[4] = src[1]; dst[5] = src[2]; dst[6] = src[3];
}

/* common/gapped_candidate.c:71-93 normalise_pcan_sdir: express the pair with the subject on its forward strand */
void ora_normalise_pcan_sdir(uint32_t item[7], uint32_t qsize, uint32_t ssize)
{
    if (!(item[0] >> 31)) return;
    uint32_t flags = item[0] & ((1u << 29) - 1);
    if (!((item[0] >> 30) & 1)) flags |= 1u << 30;     /* query strand flips */
    if (!(item[0] & (1u << 29))) flags |= 1u << 29;    /* the anchor moves to the other chain end */
    item[0] = flags;
    const uint32_t qb = qsize - item[6], qe = qsize - item[5];
    item[5] = qb; item[6] = qe;
    const uint32_t sb = ssize - item[3], se = ssize - item[2];
    item[2] = sb; item[3] = se;
}

/* common/gapped_candidate.c:95-121 PackedGappedCandidate_CnsScoreGT; ties of the reference's comparator are
 * broken by the remaining words so that the order is total (the reference's introsort leaves them unspecified) */
static int cns_score_cmp(const void* pa, const void* pb)
{
    const uint32_t* a = (const uint32_t*)pa; const uint32_t* b = (const uint32_t*)pb;
    const int sa = (int)(a[0] & ((1u << 29) - 1)), sb = (int)(b[0] & ((1u << 29) - 1));
    if (sa != sb) return sa > sb ? -1 : 1;
    if ((int)a[4] != (int)b[4]) return (int)a[4] < (int)b[4] ? -1 : 1;
    const int da = (int)(a[0] >> 30) & 1, db = (int)(b[0] >> 30) & 1;
    if (da != db) return da < db ? -1 : 1;
    if ((int)a[5] != (int)b[5]) return (int)a[5] < (int)b[5] ? -1 : 1;
    if ((int)a[2] != (int)b[2]) return (int)a[2] < (int)b[2] ? -1 : 1;
    for (int k = 0; k < 7; ++k) if (a[k] != b[k]) return a[k] < b[k] ? -1 : 1;
    return 0;
}
void ora_cns_sort_candidates(uint32_t* items, size_t n) { qsort(items, n, 28, cns_score_cmp); }

static int sid_cmp(const void* pa, const void* pb)
{
    const uint32_t* a = (const uint32_t*)pa; const uint32_t* b = (const uint32_t*)pb;
    if ((int)a[1] != (int)b[1]) return (int)a[1] < (int)b[1] ? -1 : 1;
    return cns_score_cmp(pa, pb);
}

/* ---- the rules of the loop ---- */

/* consensus/consensus_aux.c:92-113 */
static int full_cov_ovlp(int ql, int qr, int qs, int tl, int tr, int ts, int L, int M)
{
    if (ql <= M && qs - qr <= M) return 1;
    if (tl <= M && ts - tr <= M) return 1;
    if (qs - qr <= M) { if (tl > M) return 0; if (qr - ql >= L) return 1; }
    if (ts - tr <= M) { if (ql > M) return 0; if (qr - ql >= L) return 1; }
    return 0;
}

/* consensus/consensus_aux.c:115-122 */
static int mapping_range_ok(int ql, int qr, int qs, int tl, int tr, int ts, int min_ovlp, double ratio)
{
    if (qr - ql >= min_ovlp || tr - tl >= min_ovlp) return 1;
    return (qr - ql >= qs * ratio) || (tr - tl >= ts * ratio);
}

/* consensus/error_estimate.c:7-29 */
static int good_overlap(int qoff, int qend, int qsize, int soff, int send, int ssize)
{
    const int M = 200, qlh = qoff, qrh = qsize - qend, slh = soff, srh = ssize - send;
    return (qlh <= M && qrh <= M) || (slh <= M && srh <= M) || (qrh <= M && slh <= M) || (srh <= M && qlh <= M);
}

/* consensus/consensus_one_read.c:11-16 calc_cns_weight */
static double cns_weight(double ident_perc)
{
    const double e = (100.0 - ident_perc) / 100.0 / 2.0;
    double w = (1.0 - e) * (1.0 - e) + e * e / 3.0;
    if (100.0 - ident_perc <= 1.0e-6) w = 1.0;
    return w;
}

/* consensus/consensus_one_read.c:145-151 region_coverage_is_full */
static int region_full(const int* cov, int from, int to, int max_cov)
{
    for (int i = from; i < to; ++i) if (cov[i] < max_cov) return 0;
    return 1;
}

static int dbl_desc(const void* a, const void* b)
{
    const double x = *(const double*)a, y = *(const double*)b;
    return x > y ? -1 : (x < y ? 1 : 0);
}

/* consensus/error_estimate.c:31-63 estimate_ident_lower_bound */
static double ident_lower_bound(const double* ident, int n)
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

typedef struct {   /* OverlapIndex, consensus/overlaps_pool.h:9-19 */
    int cand, qid, qsize, qoff, qend, toff, tend, align_size;
    double ident_perc;
    size_t str_at;
} pool_item;

static void res_reserve_str(ora_cns_result* r, size_t more)
{
    if (r->n_strs + more <= r->m_strs) return;
    r->m_strs = (r->n_strs + more) * 2 + 4096;
    r->strs = (char*)realloc(r->strs, r->m_strs);
}

static void res_push_overlap(ora_cns_result* r, int cand, const ora_align_result* a, double ident, double weight,
                             const char* q, const char* t, int n)
{
    if (r->n_overlaps == r->m_overlaps) {
        r->m_overlaps = r->m_overlaps ? r->m_overlaps * 2 : 256;
        r->overlaps = (ora_cns_overlap*)realloc(r->overlaps, r->m_overlaps * sizeof *r->overlaps);
    }
    ora_cns_overlap* o = &r->overlaps[r->n_overlaps++];
    o->cand = cand; o->qoff = a->qoff; o->qend = a->qend; o->toff = a->toff; o->tend = a->tend;
    o->align_size = n; o->ident_perc = ident; o->weight = weight;
    res_reserve_str(r, 2 * (size_t)n);
    o->str_at = r->n_strs;
    memcpy(r->strs + r->n_strs, q, (size_t)n); memcpy(r->strs + r->n_strs + n, t, (size_t)n);
    r->n_strs += 2 * (size_t)n;
}

static void res_push_range(ora_cns_result* r, int a, int b)
{
    if (r->n_ranges == r->m_ranges) {
        r->m_ranges = r->m_ranges ? r->m_ranges * 2 : 64;
        r->ranges = (int*)realloc(r->ranges, r->m_ranges * 2 * sizeof(int));
    }
    r->ranges[2 * r->n_ranges] = a; r->ranges[2 * r->n_ranges + 1] = b; ++r->n_ranges;
}

void ora_cns_result_free(ora_cns_result* r)
{
    free(r->templates); free(r->overlaps); free(r->ranges); free(r->strs);
    memset(r, 0, sizeof *r);
}

/*
 * consensus_one_read (consensus/consensus_one_read.c:221-372) up to the hand-over to the consensus:
 * cands[0..n) = the template's candidates in examination order (already sorted by CnsScoreGT and cut to
 * MAX_EXAMINED_CAN = 300; n_all = their number before the cut), subject forward, ids global in `reads`.
 */
void ora_cns_extension_loop(const ora_volume* reads, const ora_candidate* cands, size_t n, size_t n_all,
                            const ora_cns_options* opt, ora_aligner* al, ora_cns_result* res)
{
    if (res->n_templates == res->m_templates) {
        res->m_templates = res->m_templates ? res->m_templates * 2 : 64;
        res->templates = (ora_cns_template*)realloc(res->templates, res->m_templates * sizeof *res->templates);
    }
    ora_cns_template* T = &res->templates[res->n_templates++];
    memset(T, 0, sizeof *T);
    T->ovlp_begin = T->ovlp_end = res->n_overlaps;
    T->range_begin = T->range_end = res->n_ranges;
    if (n == 0) return;
    T->template_id = cands[0].sid;
    const int tsize = (int)reads->size[cands[0].sid];
    T->template_size = tsize;
    if ((size_t)opt->min_cov > n_all) return;                       /* :223 */
    T->examined = 1;

    uint8_t* target = (uint8_t*)malloc((size_t)tsize + 1);
    ora_volume_extract(reads, (uint64_t)cands[0].sid, 0, target);
    int* cov = (int*)calloc((size_t)tsize + 1, sizeof(int));
    int* ext_ids = (int*)malloc((n + 1) * sizeof(int));              /* ReadIdPool: a set of read ids */
    size_t n_ext = 0;
    uint8_t* query = NULL; size_t query_cap = 0;
    #define EXTENDED(id) ({ int f_ = 0; for (size_t z_ = 0; z_ < n_ext; ++z_) if (ext_ids[z_] == (id)) { f_ = 1; break; } f_; })

    double ident_cutoff;
    size_t next = 0;
    int num_can = 0, num_ovlps = 0;
    ora_align_result ar;

    if (opt->use_fixed_ident_cutoff) {                               /* :267-272 */
        ident_cutoff = 100.0 * (1.0 - opt->error);
    } else {
        /* get_good_overlaps, consensus/error_estimate.c:96-183: align the first candidates (at most 50) until 15
         * of them are end-to-end overlaps, then derive the identity cutoff from the identities seen */
        enum { NIdent = 15 };
        double ident[NIdent];
        int n_ident = 0;
        pool_item* pool = (pool_item*)malloc(64 * sizeof(pool_item));
        size_t n_pool = 0;
        char* pstr = NULL; size_t n_pstr = 0, m_pstr = 0;
        size_t i;
        for (i = 0; i < n && i < 50; ++i) {
            const ora_candidate* c = &cands[i];
            if (EXTENDED(c->qid)) continue;
            const int qsize = (int)reads->size[c->qid];
            if ((size_t)qsize + 1 > query_cap) { query_cap = (size_t)qsize * 2 + 1; query = (uint8_t*)realloc(query, query_cap); }
            ora_volume_extract(reads, (uint64_t)c->qid, c->qdir, query);
            if (!ora_onc_align(al, query, (int)c->qoff, qsize, target, (int)c->soff, tsize, 512, opt->min_align_size, 4, &ar)) continue;
            pool_item* p = &pool[n_pool++];
            p->cand = (int)i; p->qid = c->qid; p->qsize = qsize; p->qoff = ar.qoff; p->qend = ar.qend; p->toff = ar.toff; p->tend = ar.tend;
            p->align_size = ar.align_size; p->ident_perc = ar.ident_perc; p->str_at = n_pstr;
            if (n_pstr + 2 * (size_t)ar.align_size > m_pstr) { m_pstr = (n_pstr + 2 * (size_t)ar.align_size) * 2; pstr = (char*)realloc(pstr, m_pstr); }
            memcpy(pstr + n_pstr, ar.query_align, (size_t)ar.align_size);
            memcpy(pstr + n_pstr + ar.align_size, ar.target_align, (size_t)ar.align_size);
            n_pstr += 2 * (size_t)ar.align_size;
            ext_ids[n_ext++] = c->qid;
            if (good_overlap(ar.qoff, ar.qend, qsize, ar.toff, ar.tend, tsize)) {
                ident[n_ident++] = ar.ident_perc;
                if (n_ident == NIdent) break;                        /* i stays on this candidate */
            }
        }
        const size_t last_extended = i - 1;                          /* error_estimate.c:178 (wraps to -1 when i = 0) */
        if (n_ident < NIdent) {                                      /* get_idents, error_estimate.c:65-94 */
            int k = 0;
            for (size_t j = 0; j < n_pool && k < NIdent; ++j)
                if (good_overlap(pool[j].qoff, pool[j].qend, pool[j].qsize, pool[j].toff, pool[j].tend, tsize)) ident[k++] = pool[j].ident_perc;
            if (k < NIdent) {
                k = 0;
                for (size_t j = 0; j < n_pool && k < NIdent; ++j)
                    if (pool[j].qend - pool[j].qoff >= pool[j].qsize * 0.6 || pool[j].tend - pool[j].toff >= tsize * 0.6) ident[k++] = pool[j].ident_perc;
            }
            n_ident = k;
        }
        qsort(ident, (size_t)n_ident, sizeof(double), dbl_desc);
        if (n_ident >= 8) n_ident = (int)(n_ident * 0.7);
        ident_cutoff = ident_lower_bound(ident, n_ident);
        /* add_extended_overlaps, consensus/consensus_one_read.c:153-190 */
        for (size_t j = 0; j < n_pool; ++j) {
            const pool_item* p = &pool[j];
            if (p->ident_perc < ident_cutoff) continue;
            if (!mapping_range_ok(p->qoff, p->qend, p->qsize, p->toff, p->tend, tsize, opt->min_align_size, opt->mapping_ratio)) continue;
            ++num_ovlps;
            ora_align_result a2; a2.qoff = p->qoff; a2.qend = p->qend; a2.toff = p->toff; a2.tend = p->tend;
            res_push_overlap(res, p->cand, &a2, p->ident_perc, cns_weight(p->ident_perc), pstr + p->str_at, pstr + p->str_at + p->align_size, p->align_size);
            for (int x = p->toff; x < p->tend; ++x) ++cov[x];
            if (!EXTENDED(p->qid)) ext_ids[n_ext++] = p->qid;
            if (full_cov_ovlp(p->qoff, p->qend, p->qsize, p->toff, p->tend, tsize, 1000, 200)) res_push_range(res, p->toff, p->tend);
        }
        next = last_extended + 1;
        num_can = (int)next;
        free(pool); free(pstr);
    }

    /* consensus/consensus_one_read.c:317-372: groups of 50 candidates until the template is covered max_cov deep */
    while (next < n) {
        if (region_full(cov, 0, tsize, opt->max_cov)) break;
        const size_t from = next, to = from + 50 < n ? from + 50 : n;
        next = to;
        for (size_t i = from; i < to; ++i) {
            const ora_candidate* c = &cands[i];
            if (EXTENDED(c->qid)) continue;
            if (region_full(cov, (int)c->sbeg, (int)c->send, opt->max_cov)) continue;
            const int qsize = (int)reads->size[c->qid];
            if ((size_t)qsize + 1 > query_cap) { query_cap = (size_t)qsize * 2 + 1; query = (uint8_t*)realloc(query, query_cap); }
            ora_volume_extract(reads, (uint64_t)c->qid, c->qdir, query);
            const int ok = ora_onc_align(al, query, (int)c->qoff, qsize, target, (int)c->soff, tsize, 512, opt->min_align_size, 4, &ar);
            ++num_can;
            if (!ok) continue;
            if (ar.ident_perc < ident_cutoff && !full_cov_ovlp(ar.qoff, ar.qend, qsize, ar.toff, ar.tend, tsize, 5000, 100)) continue;
            if (!mapping_range_ok(ar.qoff, ar.qend, qsize, ar.toff, ar.tend, tsize, opt->min_align_size, opt->mapping_ratio)) continue;
            ++num_ovlps;
            res_push_overlap(res, (int)i, &ar, ar.ident_perc, cns_weight(ar.ident_perc), ar.query_align, ar.target_align, ar.align_size);
            for (int x = ar.toff; x < ar.tend; ++x) ++cov[x];
            ext_ids[n_ext++] = c->qid;
        }
    }
    #undef EXTENDED
    T->ident_cutoff = ident_cutoff; T->num_can = num_can; T->num_ovlps = num_ovlps;
    T->ovlp_end = res->n_overlaps; T->range_end = res->n_ranges;
    free(target); free(cov); free(ext_ids); free(query);
}

/* ---- a whole work directory: what oc2cns does before / around the loop ---- */

/* common/makedb_aux.c:137-153 merge_volumes: all volumes as one read set, ids in volume order */
int ora_volumes_merge(const char* wrk_dir, ora_volume* out)
{
    ora_volumes_info vi;
    if (ora_volumes_info_load(wrk_dir, &vi)) return -1;
    memset(out, 0, sizeof *out);
    uint64_t nseq = 0, nbases = 0;
    ora_volume* vols = (ora_volume*)calloc((size_t)vi.num_volumes, sizeof(ora_volume));
    for (int v = 0; v < vi.num_volumes; ++v) {
        if (ora_volume_load(vi.names[v], &vols[v])) return -1;
        nseq += vols[v].nseq; nbases += vols[v].nbases;
    }
    out->nseq = nseq; out->nbases = nbases;
    out->pac = (uint8_t*)calloc((size_t)(nbases + 3) / 4 + 1, 1);
    out->offset = (uint64_t*)malloc(nseq * 8); out->size = (uint64_t*)malloc(nseq * 8); out->hdr_offset = (uint64_t*)calloc(nseq, 8);
    uint64_t at = 0, id = 0;
    uint8_t* tmp = NULL; size_t tmp_cap = 0;
    for (int v = 0; v < vi.num_volumes; ++v) {
        for (uint64_t i = 0; i < vols[v].nseq; ++i, ++id) {
            const uint64_t sz = vols[v].size[i];
            if (sz + 1 > tmp_cap) { tmp_cap = sz * 2 + 1; tmp = (uint8_t*)realloc(tmp, tmp_cap); }
            ora_volume_extract(&vols[v], i, 0, tmp);
            out->offset[id] = at; out->size[id] = sz;
            for (uint64_t k = 0; k < sz; ++k) out->pac[(at + k) >> 2] |= (uint8_t)(tmp[k] << (2 * (3 - ((at + k) & 3))));
            at += sz;
        }
        ora_volume_free(&vols[v]);
    }
    free(tmp); free(vols);
    ora_volumes_info_free(&vi);
    return 0;
}

/* consensus/consensus_one_partition.c:10-96 (load, sort by template, normalise) + :98-108 (one template at a
 * time) + consensus_one_read.c:250-260 (order and cut of a template's candidates).  items = n packed records
 * of one partition file; they are reordered in place. */
void ora_cns_partition(const ora_volume* reads, uint32_t* items, size_t n, const ora_cns_options* opt, ora_cns_result* res)
{
    qsort(items, n, 28, sid_cmp);
    for (size_t i = 0; i < n; ++i) {
        uint32_t* it = items + 7 * i;
        ora_normalise_pcan_sdir(it, (uint32_t)reads->size[it[4]], (uint32_t)reads->size[it[1]]);
    }
    ora_aligner* al = ora_aligner_new(opt->error);
    ora_candidate* cands = (ora_candidate*)malloc(300 * sizeof(ora_candidate));
    size_t i = 0;
    while (i < n) {
        size_t j = i + 1;
        while (j < n && items[7 * j + 1] == items[7 * i + 1]) ++j;
        ora_cns_sort_candidates(items + 7 * i, j - i);             /* normalisation changes the sort keys */
        const size_t n_all = j - i, m = n_all > 300 ? 300 : n_all;
        for (size_t k = 0; k < m; ++k) {
            ora_unpack_candidate(items + 7 * (i + k), &cands[k]);
            cands[k].qsize = (int64_t)reads->size[cands[k].qid];
            cands[k].ssize = (int64_t)reads->size[cands[k].sid];
        }
        ora_cns_extension_loop(reads, cands, m, n_all, opt, al, res);
        i = j;
    }
    free(cands);
    ora_aligner_free(al);
}

unsigned long long ora_fnv64(const char* s, size_t n)
{
    unsigned long long h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)s[i]; h *= 1099511628211ULL; }
    return h;
}

static const char kLetters[] = "ACGT-";
/* the log format of oracle/cns_ref_harness.c, so that the two can be compared as text */
void ora_cns_write_log(const ora_cns_result* r, FILE* out, int full)
{
    char* q = NULL; size_t cap = 0;
    for (size_t t = 0; t < r->n_templates; ++t) {
        const ora_cns_template* T = &r->templates[t];
        if (!T->examined) continue;
        for (size_t k = T->ovlp_begin; k < T->ovlp_end; ++k) {
            const ora_cns_overlap* o = &r->overlaps[k];
            const size_t n = (size_t)o->align_size;
            fprintf(out, "A\t%d\t%d\t%.17g\t%zu\t%016llx\t%016llx", o->toff, o->tend, o->weight, n,
                    ora_fnv64(r->strs + o->str_at, n), ora_fnv64(r->strs + o->str_at + n, n));
            if (full) {
                if (2 * n + 2 > cap) { cap = 4 * n + 2; q = (char*)realloc(q, cap); }
                memcpy(q, r->strs + o->str_at, n); q[n] = '\t'; memcpy(q + n + 1, r->strs + o->str_at + n, n); q[2 * n + 1] = 0;
                fprintf(out, "\t%s", q);
            }
            fputc('\n', out);
        }
        fprintf(out, "T\t%d\t%d\t%.17g\t%d\t%d\t%zu", T->template_id, T->template_size, T->ident_cutoff, T->num_can, T->num_ovlps,
                T->range_end - T->range_begin);
        for (size_t k = T->range_begin; k < T->range_end; ++k) fprintf(out, "\t%d\t%d", r->ranges[2 * k], r->ranges[2 * k + 1]);
        fputc('\n', out);
    }
    free(q);
    (void)kLetters;
}

/* the whole stage over a work directory + the partition files oc2pcan left at `can_prefix`; 0 = ok */
int ora_cns_run(const char* wrk_dir, const char* can_prefix, const ora_cns_options* opt, const char* log_path, int full)
{
    ora_volume reads;
    if (ora_volumes_merge(wrk_dir, &reads)) return -1;
    char path[4096];
    snprintf(path, sizeof path, "%s.partitions", can_prefix);        /* partition_candidates/pcan_aux.c:28-40 */
    FILE* f = fopen(path, "r");
    int np = 0;
    if (!f || fscanf(f, "%d", &np) != 1) { if (f) fclose(f); return -2; }
    fclose(f);
    FILE* out = fopen(log_path, "w");
    if (!out) return -3;
    for (int p = 0; p < np; ++p) {
        snprintf(path, sizeof path, "%s.p%d", can_prefix, p);       /* pcan_aux.c:9-16 */
        f = fopen(path, "rb");
        if (!f) continue;
        fseek(f, 0, SEEK_END); const long bytes = ftell(f); fseek(f, 0, SEEK_SET);
        const size_t n = (size_t)bytes / 28;
        uint32_t* items = (uint32_t*)malloc(n * 28 + 28);
        if (fread(items, 28, n, f) != n) { fclose(f); free(items); fclose(out); return -4; }
        fclose(f);
        ora_cns_result res; memset(&res, 0, sizeof res);
        ora_cns_partition(&reads, items, n, opt, &res);
        ora_cns_write_log(&res, out, full);
        ora_cns_result_free(&res);
        free(items);
    }
    fclose(out);
    ora_volume_free(&reads);
    return 0;
}
