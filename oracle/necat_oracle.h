/*
 * necat_oracle.h - CPU restatement of NECAT's overlap hot path (oc2pmov).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under necat_amd/ (the product) may include, link or call
 * this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker.  It is a from-scratch plain-C restatement of the reference algorithm, each function
 * citing the reference file:line it follows; it is pinned against the reference's own code
 * compiled from /root/reference (oracle/_ref, see Makefile) by tests/test_oracle_vs_ref.py.
 */
#ifndef NECAT_ORACLE_H
#define NECAT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- volumes: common/packed_db.h:12-29, packed_db.c:317 ---- */
typedef struct {
    uint8_t*  pac;        /* 2-bit bases, first base of a byte in its top two bits */
    uint64_t  nbases;
    uint64_t  nseq;
    uint64_t* offset;     /* [nseq] */
    uint64_t* size;       /* [nseq] */
    uint64_t* hdr_offset; /* [nseq] */
    char*     hdr;        /* NUL separated names */
    uint64_t  hdr_bytes;
} ora_volume;

int  ora_volume_load(const char* path, ora_volume* v);   /* 0 = ok */
void ora_volume_free(ora_volume* v);
/* packed_db.c:255 pdb_extract_subsequence: strand 0 = FWD, 1 = REV (reverse + 3-c) */
void ora_volume_extract(const ora_volume* v, uint64_t i, int strand, uint8_t* out);
uint64_t ora_offset_to_id(const ora_volume* v, uint64_t offset); /* packed_db.c:173 */

typedef struct {
    int       num_volumes;
    int       num_reads;
    char**    names;        /* [num_volumes] */
    int*      read_start_id;
    int*      read_count;
} ora_volumes_info;
int  ora_volumes_info_load(const char* wrk_dir, ora_volumes_info* vi); /* makedb_aux.c:78 */
void ora_volumes_info_free(ora_volumes_info* vi);

/* ---- options: common/map_options.h:10-25 ---- */
typedef struct {
    int    kmer_size, scan_window, kmer_cnt_cutoff, block_size, block_score_cutoff;
    int    num_candidates, align_size_cutoff;
    double ddfs_cutoff, error;
    int    num_output, num_threads, job, binary_output, use_hdr_as_id;
} ora_options;
void ora_options_default(ora_options* o);                       /* map_options.c:12-28 */
int  ora_options_parse(int argc, char** argv, ora_options* o);  /* map_options.c:90; 0 = ok */

/* ---- k-mer index: lookup_table/lookup_table.h:6-21 ---- */
typedef struct {
    uint64_t* kmer_stats;   /* [4^k]   cnt<<34 | start */
    uint64_t* offset_list;  /* [n]     global base offsets, ascending inside one k-mer */
    uint64_t  n_offsets;
    int       k;
} ora_index;
ora_index* ora_index_build(const ora_volume* ref, int k, int max_occ); /* lookup_table.c:149 */
void       ora_index_free(ora_index* ix);

/* ---- candidates: common/gapped_candidate.h:9-19 ---- */
typedef struct {
    int32_t qid, sid, qdir, sdir, score;
    int64_t qbeg, qend, qsize, sbeg, send, ssize, qoff, soff;
} ora_candidate;

typedef struct { ora_candidate* a; size_t n, m; } ora_can_vec;

typedef struct ora_wfd ora_wfd;   /* per-worker seeding scratch (word_finder.h:11-19) */
ora_wfd* ora_wfd_new(uint64_t reference_bases, int block_size, int kmer_size, int block_score_cutoff);
void     ora_wfd_free(ora_wfd* w);
/* word_finder.c:364 find_candidates: appends candidates with LOCAL ids */
void ora_find_candidates(const uint8_t* read, int read_size, int qid, int qdir,
                         int read_start_id, int reference_start_id, int pairwise,
                         const ora_volume* reference, const ora_index* ix,
                         const ora_options* opt, ora_wfd* w, ora_can_vec* out);

/* ---- extension: gapped_align/edlib_ex.c:733, oc_aligner.c:303 ---- */
typedef struct ora_aligner ora_aligner;
ora_aligner* ora_aligner_new(double error);
void         ora_aligner_free(ora_aligner* a);
/* Edlib_align: returns 1 on success; qaln/taln receive NUL-terminated gapped strings */
int ora_edlib_align(ora_aligner* a, const uint8_t* query, int qn, const uint8_t* target, int tn,
                    char* qaln, char* taln, int* qend, int* tend, int* edit_distance);
typedef struct {
    int    qoff, qend, toff, tend;
    double ident_perc;
    int    align_size;     /* gapped columns */
    const char* query_align;   /* valid until the next call on the same aligner */
    const char* target_align;
} ora_align_result;
/* onc_align: returns 1 iff align_size >= min_align_size */
int ora_onc_align(ora_aligner* a, const uint8_t* query, int query_start, int query_size,
                  const uint8_t* target, int target_start, int target_size,
                  int block_size, int min_align_size, int tail_match_len, ora_align_result* res);

/* ---- records ---- */
typedef struct {   /* common/m4_record.h:10-25 (96 bytes, same field order) */
    int32_t qid, qdir; uint64_t qoff, qend, qext, qsize;
    int32_t sid, sdir; uint64_t soff, send, sext, ssize;
    double ident_perc; int32_t vscore; int32_t _pad;
} ora_m4;
void ora_pack_candidate(const ora_candidate* c, uint32_t item[7]); /* gapped_candidate.c:13 */

/* ---- per-read driver and whole stage: pm_worker.c:85, :338 ---- */
typedef struct {
    uint64_t n_records;       /* output records */
    uint64_t aligned_qbases;  /* sum(qend-qoff) over M4 records (job 1) */
    double   t_index, t_map;  /* seconds */
} ora_stats;
int ora_pm_main(const ora_options* opt, int vid, const char* wrk_dir, const char* output,
                ora_stats* stats);


/* ---- consensus stage, extension loop only (cns_oracle.c) ---- */
#include <stdio.h>
typedef struct {   /* the fields of consensus/cns_options.h:6-18 the loop reads */
    int    min_align_size, min_cov, max_cov;
    double error, mapping_ratio;
    int    use_fixed_ident_cutoff;
} ora_cns_options;
typedef struct {   /* one add_one_align call (tasc/cbcns.c:47) */
    int    cand;             /* index into the template's candidate list */
    int    qoff, qend, toff, tend, align_size;
    double ident_perc, weight;
    size_t str_at;           /* strs[str_at .. +align_size) = query string, then the target string */
} ora_cns_overlap;
typedef struct {
    int    template_id, template_size, examined;   /* examined = 0: fewer than min_cov candidates */
    double ident_cutoff;
    int    num_can, num_ovlps;                     /* CnsSeq fields, common/cns_seq.h:12-14 */
    size_t ovlp_begin, ovlp_end, range_begin, range_end;
} ora_cns_template;
typedef struct {
    ora_cns_template* templates; size_t n_templates, m_templates;
    ora_cns_overlap*  overlaps;  size_t n_overlaps, m_overlaps;
    int*              ranges;    size_t n_ranges, m_ranges;      /* cov_ranges pairs */
    char*             strs;      size_t n_strs, m_strs;
} ora_cns_result;
void ora_unpack_candidate(const uint32_t item[7], ora_candidate* c);
void ora_change_pcan_roles(const uint32_t src[7], uint32_t dst[7]);
void ora_normalise_pcan_sdir(uint32_t item[7], uint32_t qsize, uint32_t ssize);
void ora_cns_sort_candidates(uint32_t* items, size_t n);
void ora_cns_extension_loop(const ora_volume* reads, const ora_candidate* cands, size_t n, size_t n_all,
                            const ora_cns_options* opt, ora_aligner* al, ora_cns_result* res);
int  ora_volumes_merge(const char* wrk_dir, ora_volume* out);
void ora_cns_partition(const ora_volume* reads, uint32_t* items, size_t n, const ora_cns_options* opt, ora_cns_result* res);
void ora_cns_write_log(const ora_cns_result* r, FILE* out, int full);
int  ora_cns_run(const char* wrk_dir, const char* can_prefix, const ora_cns_options* opt, const char* log_path, int full);
void ora_cns_result_free(ora_cns_result* r);
unsigned long long ora_fnv64(const char* s, size_t n);   /* the hash of the logs (FNV-1a, 64 bit) */

#ifdef __cplusplus
}
#endif
#endif
