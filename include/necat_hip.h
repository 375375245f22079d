/*
 * necat_hip.h - C ABI of libnecat_hip.so, the MI355X (gfx950) implementation of NECAT's all-vs-all
 * overlap stage (oc2pmov: k-mer index -> seeding / DDF vote / chain DP -> block-wise banded Myers
 * extension -> candidate / M4 records).
 *
 * NECAT has no in-process plugin interface for this stage: the drop-in boundary is the
 * oc2pmov process (argv + volume files in, one record file out), which necat_amd/csrc/oc2pmov_main.cpp
 * reproduces on top of this ABI.  The entry points below mirror, at batch granularity, the three calls
 * the reference's worker (pm_one_volume/pm_worker.c) makes into libontcns, so that a maintainer could
 * also bind them directly:
 *
 *   necat_index_build      <- build_lookup_table      lookup_table/lookup_table.h:23-27, lookup_table.c:149
 *   necat_find_candidates  <- find_candidates (+ the per-read sort/truncate of pm_search_one_volume)
 *                                                     word_finder/word_finder.h:33-45, pm_worker.c:100-140,163-171
 *   necat_extend           <- extend_candidates/onc_align
 *                                                     pm_worker.c:29-83, gapped_align/oc_aligner.h:45-55
 *   necat_map_pair         <- pm_search_one_volume, -j 1: the two calls above fused (candidates stay on the device)
 *                                                     pm_worker.c:85-173
 *   necat_volume_upload    <- pdb_load                common/packed_db.c:386 (the 2-bit pac + SequenceInfo)
 *   necat_onc_align_batch  <- onc_align with its gapped strings, for the consensus client
 *                                                     gapped_align/oc_aligner.h:45-55, consensus_aux.c:124-215
 *
 * Conventions: plain C types only; every function returns 0 on success and a negative code on failure
 * (necat_last_error() gives the text); output arrays are malloc'ed by the library and released with
 * necat_free(); handles are opaque; one host thread per context.  There is no CPU fallback: every entry
 * point fails with NECAT_ERR_DEVICE if no gfx950 device is usable.
 */
#ifndef NECAT_HIP_H
#define NECAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header.  It changes whenever a struct an entry point copies into caller memory changes size or layout (round 5 grew
 * necat_timings and necat_shard_timings, round 6 necat_timings again): a caller built against another header must not pass its smaller struct to
 * necat_get_timings / necat_get_shard_timings.  Check necat_abi_version() == NECAT_ABI_VERSION once after loading the library, or use the
 * *_sized getters, which copy at most the bytes the caller says its struct has (new fields are always appended). */
#define NECAT_ABI_VERSION   6
int  necat_abi_version(void);

#define NECAT_OK            0
#define NECAT_ERR_ARG      (-1)
#define NECAT_ERR_DEVICE   (-2)   /* no usable GPU / HIP runtime error */
#define NECAT_ERR_MEMORY   (-3)
#define NECAT_ERR_CAPACITY (-4)   /* an internal device buffer overflowed (reported, never silent) */
#define NECAT_ERR_INTERNAL (-5)
#define NECAT_ERR_COMM     (-6)   /* rank-to-rank exchange failed (RCCL / HIP IPC / the host all-gather callback) */

typedef struct necat_ctx necat_ctx;
typedef struct necat_volume necat_volume;
typedef struct necat_index necat_index;
typedef struct necat_comm necat_comm;

/* common/map_options.h:10-25 (same fields, same meaning) */
typedef struct {
    int    kmer_size;           /* -k */
    int    scan_window;         /* -z */
    int    kmer_cnt_cutoff;     /* -q */
    int    block_size;          /* -b */
    int    block_score_cutoff;  /* -s */
    int    num_candidates;      /* -n */
    int    align_size_cutoff;   /* -a */
    double ddfs_cutoff;         /* -d (parsed, ignored by the reference: word_finder.c:10) */
    double error;               /* -e */
    int    num_output;          /* -m (unused by oc2pmov) */
    int    num_threads;         /* -t */
    int    job;                 /* -j 0 = candidates, 1 = align */
    int    binary_output;       /* -u */
    int    use_hdr_as_id;       /* -i */
} necat_map_options;

/* common/gapped_candidate.h:9-19 (GappedCandidate; idx fields are 64-bit unsigned there) */
typedef struct {
    int32_t  qid, sid, qdir, sdir, score;
    int32_t  _pad;
    uint64_t qbeg, qend, qsize;
    uint64_t sbeg, send, ssize;
    uint64_t qoff, soff;
} necat_candidate;

/* common/m4_record.h:10-25 (M4Record, 96 bytes, identical field order) */
typedef struct {
    int32_t  qid, qdir;
    uint64_t qoff, qend, qext, qsize;
    int32_t  sid, sdir;
    uint64_t soff, send, sext, ssize;
    double   ident_perc;
    int32_t  vscore;
    int32_t  _pad;
} necat_m4;

/* what onc_align leaves in OcAlignData (gapped_align/oc_aligner.h:6-15): strand coordinates of the
 * alignment, its length in gapped columns and identity; ok = onc_align's return value */
typedef struct {
    int32_t ok;
    int32_t qoff, qend, toff, tend;
    int32_t align_size;
    double  ident_perc;
} necat_alignment;

/* wall-clock (ms, HIP events) of the last call of each stage + work counters of the extension */
typedef struct {
    double   index_ms, seed_ms, extend_ms;
    double   myers_ms;          /* sum over launches of the block Myers DP kernel */
    double   traceback_ms;
    uint64_t myers_launches;
    uint64_t myers_blocks;      /* block alignments (Edlib_align equivalents) */
    uint64_t myers_word_updates;/* 64-row word updates actually computed (SHW + NW) */
    uint64_t myers_cells_bases; /* sum over blocks of query+target fragment bases */
    uint64_t rounds;
    /* the list-A kernels alone (blocks <= 512 x 512), launches of more than NECAT_SINGLE_PASS blocks: the two-pass
     * DP kernel k_myers_coop<8,16,512,8,false> and its traceback k_traceback<8,16,512,..> */
    double   myersA_ms;
    uint64_t myersA_launches, myersA_blocks;
    double   tracebackA_ms;
    /* the list-A DP launch with the most blocks (the throughput-bound regime of the dominant kernel) */
    double   myersA_big_ms;
    uint64_t myersA_big_blocks;
    /* band words the NW passes stored (= the word updates the reference's banded NW pass needs, edlib_ex.c:311-325 with
     * k = the block's distance): with myers_word_updates, the redundant part of the computed work */
    uint64_t myers_band_words;
    /* the small lists of the late rounds: fragments + single-pass DP + walk + plan of the next block in ONE launch per list and
     * round, the band in LDS (necat_amd/csrc/ext_tail.h); their blocks are part of myers_blocks, their time is not in myers_ms /
     * traceback_ms */
    double   fused_ms;
    uint64_t fused_launches, fused_blocks;
    /* the full 512 x 512 blocks of the big list-A rounds (necat_amd/csrc/ext_rcwalk.h): k_myers_ck (SHW pass + checkpoints; its time is
     * what myersA_ms holds for those rounds) and k_rcwalk4 (the walk that recomputes its cells: rc_ms); rc_blocks of them, rc_words word
     * updates recomputed by the walk (each block also cost 4096 in the SHW pass) */
    double   rc_ms;
    uint64_t rc_launches, rc_blocks, rc_words;
    double   rc_ck_ms;          /* k_myers_ck of those rounds (beside the ragged blocks' DP kernel on another stream) */
    /* work counters of the last seeding call (necat_find_candidates* / the seeding part of necat_map_pair*), the terms of SURVEY 8d's
     * B_seed = L / 4 + 8 lookups + 8 hits + 28 candidates: bases of the query strands walked (both strands of every read of the
     * call), sampled k-mers looked up in the table (word_finder.c:66-83, one kmer_stats word each), offset-list entries those k-mers
     * own (what collect_seeds reads, word_finder.c:107-139) and candidates emitted */
    uint64_t seed_bases, seed_lookups, seed_hits, seed_cands;
} necat_timings;

/* Threads.  A context is used by ONE host thread at a time (its streams, arenas, timings and last-error text are its own); DIFFERENT contexts -
 * of one device or of several - may be called from different threads at the same moment, and nothing else in the library is shared but the pool
 * of pinned result blocks behind necat_free (a mutex).  Volumes and indexes are plain device allocations: made through one context, they may be
 * read by calls on any other context of that device once the call that made them has returned (necat_volume_free / necat_index_free when no call
 * uses them any more).  That is how several pairs are kept in flight on one device - one context + host thread per pair, one shared index:
 * INTEGRATION.md 2g, the programs' NECAT_PAIR_LANES, bench.py --in-flight (reference analogue: pm_main's thread pool on one lookup table,
 * pm_worker.c:335-390).  necat_ctx_trim synchronises the whole DEVICE: not while another context has calls in flight. */
void        necat_default_options(necat_map_options* o);            /* map_options.c:12-28 */
int         necat_ctx_create(int device_id, necat_ctx** out);
void        necat_ctx_destroy(necat_ctx* ctx);

/* Release the context's cached device scratch (index-build partitions, seeding arenas, band pools ...); it is allocated again on
 * demand.  For short-lived processes that build an index once: on MI355X a fresh process pays tens of ms per GB of VRAM that is
 * still being cleaned after the previous one, so a command-line program keeps its peak small.  (No reference counterpart.) */
void        necat_ctx_trim(necat_ctx* ctx);
const char* necat_last_error(const necat_ctx* ctx);    /* ctx == NULL: why this thread's last necat_ctx_create failed */
int         necat_device_name(const necat_ctx* ctx, char* buf, size_t n);

/* pac: NECAT 2-bit bases (first base of a byte in its top two bits, ontcns_aux.h:118-119);
 * seq_offset/seq_size: SequenceInfo.offset/.size (packed_db.h:12-18). Host buffers are not retained. */
int  necat_volume_upload(necat_ctx* ctx, const uint8_t* pac, uint64_t nbases,
                         const uint64_t* seq_offset, const uint64_t* seq_size, uint64_t nseq,
                         necat_volume** out);
/* <- pdb_add_one_seq (packed_db.c:229-252) for a whole volume: ASCII bases (A C G T in either case; '-' and every other
 * character as common/nst_nt4_table.c codes them, OR-ed into the pac byte exactly like the reference) packed on the device.
 * pac_out (may be NULL): (nbases + 3) / 4 bytes, what oc2mkdb writes after the volume's headers.  out (may be NULL): the
 * uploaded volume, as necat_volume_upload of those bytes would give. */
int  necat_volume_pack(necat_ctx* ctx, const char* ascii, uint64_t nbases, const uint64_t* seq_offset,
                       const uint64_t* seq_size, uint64_t nseq, uint8_t* pac_out, necat_volume** out);
void necat_volume_free(necat_ctx* ctx, necat_volume* v);

/* Final LookupTable contents (lookup_table.h:6-21): kmer_stats[h] = cnt<<34 | start for k-mers with
 * 1..max_occ occurrences (0 otherwise); offset_list = base offsets grouped by hash, ascending inside
 * one hash. */
int  necat_index_build(necat_ctx* ctx, const necat_volume* ref, int kmer_size, int max_occ,
                       necat_index** out);
int  necat_index_size(const necat_index* ix, uint64_t* table_entries, uint64_t* n_offsets);
/* copy the index to host buffers (either may be NULL); used by parity tests */
int  necat_index_download(necat_ctx* ctx, const necat_index* ix, uint64_t* kmer_stats, uint64_t* offset_list);
/* The table as the device holds it when it was built in the sparse layout (k >= 11): per 64 table entries one pair of words
 * (bits: which of the 64 entries are non-zero; base: position in `compact` of the first of them), and `compact`, the non-zero
 * `cnt<<34 | start` entries in hash order - kmer_stats[h] = bit (h & 63) of bits[h >> 6] set ? compact[base[h >> 6] + popcount(bits
 * below it)] : 0.  0.27 + (8 bytes per distinct kept k-mer) GB at k = 15 instead of the 8.6 GB of the dense table: what a host-side
 * reader of the table (the vote of oc2asmpm, asm_pm_common.c:509-702) copies.  necat_index_sparse_size: n_pairs = 0 for an index in
 * the dense layout (small k: use necat_index_download).  Outputs of the download may be NULL (skipped); pairs = 2 * n_pairs words. */
int  necat_index_sparse_size(const necat_index* ix, uint64_t* n_pairs, uint64_t* n_compact);
int  necat_index_download_sparse(necat_ctx* ctx, const necat_index* ix, uint64_t* pairs, uint64_t* compact, uint64_t* offset_list);
void necat_index_free(necat_ctx* ctx, necat_index* ix);

/* All reads of `reads` (both strands) against `ref`.  Output: per read, its candidates in exactly the
 * order pm_search_one_volume would hand them on (job 0: discovery order FWD then REV, sorted by
 * GappedCandidate_PmScoreGT and cut to num_candidates only when more than num_candidates; job 1:
 * always sorted and cut), reads in ascending id; ids are GLOBAL (+= read_start_id / ref_start_id,
 * pm_worker.c:165-166).  `pairwise` as in find_candidates (self-volume: only subjects before the
 * query, word_finder.c:121-127). */
int  necat_find_candidates(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref,
                           const necat_volume* reads, int read_start_id, int ref_start_id,
                           int pairwise, const necat_map_options* opt,
                           necat_candidate** out, uint64_t* n_out);

/* onc_align on every candidate (ids global, as produced above; block size kOcaBlockSize = 512,
 * edlib_ex_aux.h:23), then the per-read containment filter of extend_candidates in candidate order
 * (pm_worker.c:44, map_aux.c:4).  Output records carry global ids; REV query coordinates are flipped
 * to forward-strand numbering (pm_worker.c:73-78). */
int  necat_extend(necat_ctx* ctx, const necat_volume* ref, const necat_volume* reads,
                  int read_start_id, int ref_start_id,
                  const necat_candidate* cands, uint64_t n, const necat_map_options* opt,
                  int tail_match_len, necat_m4** out, uint64_t* n_out);

/* pm_search_one_volume of a mapping job (pm_worker.c:85-173 with -j 1) for every read of `reads` in one call:
 * necat_find_candidates followed by necat_extend, with the candidates never leaving the device.  Same M4
 * records as the two calls made one after the other; *n_candidates (optional) = candidates examined. */
int  necat_map_pair(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                    int read_start_id, int ref_start_id, int pairwise, const necat_map_options* opt,
                    int tail_match_len, necat_m4** out, uint64_t* n_out, uint64_t* n_candidates);

/* rm_search_one_volume (reference_mapping/rm_worker.c:198-291) for every read of `reads` against the reference volume `ref`
 * (oc2rm_worker, necat.pl:661): candidates of both strands against the whole reference (pairwise = FALSE), sorted and cut to
 * num_candidates; every candidate is aligned block-wise against the stretch of its reference sequence the read can reach from the
 * anchor (calc_reference_range, :43-59; tail_match_len = ONC_TAIL_MATCH_LEN_SHORT) on the device; then, per read and in candidate
 * order on the host threads (rm_extend_candidates, :165-196): anchors inside an accepted record are skipped, failed alignments
 * dropped, alignments more than 500 bp short of the chained range replaced by the rescue pair's (ocda_go + edlib_go, :113-140) or
 * dropped with it.  Records as rm_extend_candidate leaves them (:143-163), ids global.  *n_rescued (optional) = records that came
 * from the rescue pair. */
int  necat_map_reference(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                         int read_start_id, int ref_start_id, const necat_map_options* opt,
                         necat_m4** out, uint64_t* n_out, uint64_t* n_candidates, uint64_t* n_rescued);

/* onc_align (gapped_align/oc_aligner.h:45-55) on every candidate WITH the alignment itself - the call the
 * consensus stage makes (cns_extension, consensus/consensus_aux.c:124-215, tail_match_len =
 * ONC_TAIL_MATCH_LEN_LONG = 4).  No containment filter: aln[i] and the columns at ops + ops_off[i] belong to
 * cands[i].  TWO BITS per gapped column, in alignment order, four columns per byte from the low bits up
 * (column j of an alignment = bits 2 (j & 3) of its byte j >> 2): 0 match, 1 query base over '-' in
 * target_align, 2 '-' in query_align over a target base, 3 mismatch; aln[i].align_size columns, every alignment
 * starts on an 8-byte boundary (ops_off[i] = byte offset, ops_off[n] = total bytes); necat_gapped_strings()
 * turns them into the reference's two "ACGT-" strings. */
int  necat_onc_align_batch(necat_ctx* ctx, const necat_volume* ref, const necat_volume* reads,
                           int read_start_id, int ref_start_id,
                           const necat_candidate* cands, uint64_t n, const necat_map_options* opt,
                           int tail_match_len, necat_alignment** aln, uint8_t** ops, uint64_t** ops_off);

/* blockwise_edlib_align (asm_pm/blockwise_edlib.c:1205-1371) for n anchors at once: the block aligner of oc2asmpm = onc_align with
 * 2048-bp blocks and tail match length 8 (hbn_align.c:8, blockwise_edlib.c:910-911), the read on its forward strand against the subject
 * on strand sdir (asm_pm_common.c:133-143 turns a reverse query into a reverse subject).  anchors[i]: ids global, qoff on the forward
 * read, soff on strand sdir of the subject.  Outputs as necat_onc_align_batch: aln[i] (ok = at least min_align_size columns; the
 * identity test, hbn_align.c / blockwise_edlib.c:1357, is the caller's) and the columns, two bits each, at ops + ops_off[i]; expand them
 * with necat_gapped_strings (tseq = the subject ON ITS STRAND). */
typedef struct { int32_t qid, sid, sdir, qoff, soff; } necat_asm_anchor;
int  necat_asm_align_batch(necat_ctx* ctx, const necat_volume* ref, const necat_volume* reads, int read_start_id, int ref_start_id,
                           const necat_asm_anchor* anchors, uint64_t n, double error, int min_align_size,
                           necat_alignment** aln, uint8_t** ops, uint64_t** ops_off);

/* oc2asmpm's candidate stage for all reads of a volume at once (asm_pm/asm_pm_common.c): the 1000-bp block vote of both strands of every read
 * (pairwise_mapping :509-702, find_location :479-507), the per-read order and cut (AsmGappedCandidate_ScoreGT :145-153, the first -n candidates,
 * extend_candidates :329-345) and, for every distinct (subject, strand) among those in that order, the chained range of compute_align_range_1
 * (asm_pm/find_mem.c:222-265: 10-mer matches -> maximal exact matches -> mem_find_best_can, asm_pm/km_chain.c:322-445).  opt: kmer_size (the index's),
 * scan_window (BC), num_candidates.  out[first[r] .. first[r + 1]): read r's entries in the walk's order; ids global; qoff (forward read) / soff (on
 * strand sdir of the subject) = the anchor, score = the chain's score; qoff < 0: no chain of score >= 100, nothing to align.  Both arrays: necat_free. */
typedef struct { int32_t qid, sid, sdir, qoff, soff, score, ssize; } necat_asm_plan;
int  necat_asm_plan_batch(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads, int read_start_id, int ref_start_id,
                          const necat_map_options* opt, necat_asm_plan** out, uint64_t** first);

/* Host helper: expand `n` packed columns into query_align / target_align (each n bytes, no terminator).
 * qseq / tseq: byte codes 0..3 of the query STRAND (reverse complement for qdir = 1) and of the subject;
 * qoff / toff: the alignment's start in them (necat_alignment.qoff / .toff).  Returns 0, or NECAT_ERR_ARG
 * if the columns run past qsize / tsize. */
int  necat_gapped_strings(const uint8_t* ops, uint64_t n, const uint8_t* qseq, uint64_t qsize, uint64_t qoff,
                          const uint8_t* tseq, uint64_t tsize, uint64_t toff, char* query_align, char* target_align);

/* ---- consensus stage (oc2cns), extension loop only - SURVEY 8f.1 ------------------------------------------
 * What consensus_one_read (consensus/consensus_one_read.c:221-372) does for ONE template before it hands
 * over to the consensus proper (tasc/cbcns.c): align its candidates in score order - first until 15
 * end-to-end overlaps give an identity cutoff (get_good_overlaps, consensus/error_estimate.c:96-183), then in
 * groups of 50 until the template is covered max_cov deep - and pass every accepted alignment to
 * add_one_align (tasc/cbcns.c:47) with a weight.  necat_cns_extension_batch does that for MANY templates in
 * one call: the candidates each template's loop would reach next are aligned speculatively, all templates
 * together, on the device (the DP kernels of necat_onc_align_batch, tail_match_len = 4); the loop's
 * sequential decisions (read already used, region already covered, cutoff) are then replayed in order on
 * the results, so the overlaps, their order and the per-template numbers are those of the sequential loop.
 * With rescue_long_indels (-r 1; default 0: cns_options.c:19) a candidate whose block-wise extension failed or stopped more than
 * 200 bp short of its chained range is aligned again on the HOST, as in the reference (consensus_aux.c:168-199): DALIGNER's local
 * alignment around the anchor, then edlib's global path over that range (necat_amd/csrc/rescue.h, cns_rescue.h) - after each
 * device pass, for the candidates that need it, on all host threads. */

/* the fields of CnsOptions (consensus/cns_options.h:6-18) the loop reads; defaults cns_options.c:10-22 */
typedef struct {
    int    min_align_size;          /* -a 400 */
    int    min_cov;                 /* -x 4   */
    int    max_cov;                 /* -y 12  */
    double error;                   /* -e 0.5 */
    double mapping_ratio;           /* -p 0.8 */
    int    use_fixed_ident_cutoff;  /* -u 0   */
    int    rescue_long_indels;      /* -r 0   */
} necat_cns_options;
void necat_cns_default_options(necat_cns_options* o);

/* one add_one_align call: the overlap of candidate `cand` with its template */
typedef struct {
    uint64_t cand;            /* index into the cands array of the call */
    int32_t  qoff, qend, toff, tend;   /* as in necat_alignment */
    int32_t  align_size;      /* gapped columns */
    uint32_t ops_block;       /* the columns start at result->ops[ops_block] + ops_off (bytes), align_size of them, */
    uint64_t ops_off;         /* packed 2 bits per column as in necat_onc_align_batch                               */
    double   ident_perc;
    double   weight;          /* calc_cns_weight, consensus_one_read.c:11-16 */
} necat_cns_overlap;

/* what consensus_one_read leaves for one template */
typedef struct {
    int32_t  examined;        /* 0: fewer than min_cov candidates, the template is skipped (:223) */
    int32_t  num_can;         /* CnsSeq.num_can   (common/cns_seq.h:12) */
    int32_t  num_ovlps;       /* CnsSeq.num_ovlps (:13) = ovlp_end - ovlp_begin */
    int32_t  _pad;
    double   ident_cutoff;    /* CnsSeq.ident_cutoff (:14) */
    uint64_t ovlp_begin, ovlp_end;     /* its add_one_align calls, in call order: overlaps[ovlp_begin .. ovlp_end) */
    uint64_t range_begin, range_end;   /* its cov_ranges: pairs ranges[2 i], ranges[2 i + 1] */
} necat_cns_template;

typedef struct {
    uint64_t            n_templates;
    necat_cns_template* templates;
    uint64_t            n_overlaps;
    necat_cns_overlap*  overlaps;
    uint64_t            n_ranges;
    int32_t*            ranges;
    uint32_t            n_ops_blocks;
    uint8_t**           ops;           /* blocks of alignment columns (pinned host memory) */
    /* work counters */
    uint64_t            n_aligned;     /* alignments computed, speculative ones included */
    uint64_t            n_used;        /* alignments the sequential loop computes (cns_extension calls) */
    uint32_t            n_rounds;      /* device passes (one necat_onc_align_batch-like call each) */
    double              device_ms;     /* sum of the passes (HIP events, result copies included) */
    double              host_ms;       /* select + replay on the host */
    /* -r 1 */
    uint64_t            n_rescue_tried, n_rescued;   /* candidates handed to the host pair / alignments it replaced */
    double              rescue_ms;     /* wall time of the pair over all passes (not part of host_ms) */
} necat_cns_result;

/* Order and cut of one partition file's candidates as oc2cns does it: records of one template together
 * (load_partition_candidates, consensus/consensus_one_partition.c:10-52), subject strand normalised to
 * forward (normalise_pcan_sdir, common/gapped_candidate.c:71-93), each template's candidates sorted by
 * PackedGappedCandidate_CnsScoreGT (gapped_candidate.c:95-121) and cut to MAX_EXAMINED_CAN = 300
 * (consensus_aux.h:15, consensus_one_read.c:250-260).  packed = n 28-byte PackedGappedCandidate records with
 * ids global in `reads`.  Outputs (necat_free each): cands; tmpl_off[n_templates + 1] (template t owns
 * cands[tmpl_off[t] .. tmpl_off[t+1])); n_all[n_templates] = candidates of the template before the cut. */
int  necat_cns_load_partition(necat_ctx* ctx, const necat_volume* reads, const void* packed, uint64_t n,
                              necat_candidate** cands, uint64_t** tmpl_off, uint64_t** n_all, uint64_t* n_templates);

/* cands: per template in examination order (as necat_cns_load_partition leaves them): sid = the template,
 * sdir = 0, ids global in `reads` (the merged read set, common/makedb_aux.c:137), qsize / ssize filled.
 * n_all may be NULL (= the counts given). */
int  necat_cns_extension_batch(necat_ctx* ctx, const necat_volume* reads, const necat_candidate* cands,
                               const uint64_t* tmpl_off, const uint64_t* n_all, uint64_t n_templates,
                               const necat_cns_options* opt, necat_cns_result** out);
void necat_cns_result_free(necat_cns_result* r);

/* ---- the candidate partitioner of the consensus stage ------------------------------------------
 * <- partition_candidates/pcan.c:39-103 for candidates that are still in this process (what necat_find_candidates just
 * returned), instead of the write + read of the candidates file: every candidate is offered as it is (template = its subject)
 * and with the roles exchanged (change_pcan_roles, common/gapped_candidate.c:54-69); partition i holds the records whose
 * template id lies in [i * batch_size, (i + 1) * batch_size), i < num_parts = ceil(num_reads / batch_size).  Outputs (necat_free
 * both): records = 28-byte PackedGappedCandidate records (7 x uint32), grouped by partition, in no particular order inside one
 * (as in the reference); part_off[num_parts + 1] in records. */
int  necat_pcan_partition(necat_ctx* ctx, const necat_candidate* cands, uint64_t n, int batch_size, int num_reads,
                          uint32_t** records, uint64_t** part_off, int* num_parts);

/* ---- one reference volume on several GPUs (fine granularity) -------------------------------------
 * The reference parallelises ONE volume over threads that pull 500-read chunks from a counter
 * (pm_worker.c:13,354-362; common/map_aux.c:59-77) against one shared lookup table.  The multi-GPU equivalent:
 * one process per GPU (rank), every rank holds the volume,
 *   - the index is built in hash-range slices - rank g counts, filters and ranks only the k-mers whose hash falls
 *     in its range of the 4^k table - and the slices of the table / of offset_list are all-gathered (starts already
 *     final: shifted by the exclusive scan of the slice sizes before they are written), so every rank ends up with the
 *     COMPLETE index (necat_index_download gives the reference-layout arrays): necat_index_build_sharded;
 *   - the query reads are dealt out in chunks of `chunk_reads` reads, chunk c to rank c % nranks (reads late in the
 *     volume see more subjects - word_finder.c:121-127 - so contiguous ranges would not balance); every read is
 *     processed exactly as on one GPU, so the union of the ranks' records IS the single-GPU record set;
 *   - the records are gathered (gather-v, device to device) on rank `root`.
 * Device memory moves by RCCL send/recv groups (direct all-pairs over xGMI) or, for ranks that share a device,
 * by HIP IPC copies (transport "ipc"; "auto" picks RCCL unless two ranks sit on one device).  Small host-side
 * values (slice sizes, the RCCL id, IPC handles) travel through the caller's all-gather callback. */

/* all-gather of `bytes` bytes per rank among the job's ranks: recv = nranks * bytes, rank order; returns 0 on success */
typedef int (*necat_host_allgather_fn)(void* user, const void* send, void* recv, size_t bytes);

int  necat_comm_create(necat_ctx* ctx, int rank, int nranks, necat_host_allgather_fn fn, void* user,
                       const char* transport /* "auto" | "rccl" | "ipc" */, necat_comm** out);
void necat_comm_destroy(necat_comm* c);
/* "rccl" or "ipc" */
int  necat_comm_transport(const necat_comm* c, char* buf, size_t n);

/* wall-clock of the last sharded calls on this rank */
typedef struct {
    double   index_local_ms;     /* this rank's slice of the build (HIP events) */
    double   index_exchange_ms;  /* all-gather of the table and offset_list slices (wall) */
    uint64_t index_exchange_bytes; /* bytes this rank received */
    double   gather_ms;          /* gather-v of the records on the root (wall) */
    uint64_t gather_bytes;
    uint64_t reads_local;        /* query reads this rank processed */
    /* how the last necat_index_build_sharded built its index: 1 = hash-range slices + all-gather, 0 = every rank built the whole table (no
     * exchange), and the two costs necat_index_plan priced for it */
    uint64_t index_sharded;
    double   index_plan_replicate_ms, index_plan_shard_ms;
} necat_shard_timings;
int  necat_get_shard_timings(const necat_ctx* ctx, necat_shard_timings* t);
int  necat_get_shard_timings_sized(const necat_ctx* ctx, void* t, size_t bytes);

/* The cost model behind necat_index_build_sharded's choice (no reference counterpart: build_lookup_table, lookup_table.c:149, is one thread).
 * SURVEY 8e's sharded build was specified against a 66 s CPU build; on the device ONE rank builds an E. coli-size table in 5 ms, so slicing the build
 * pays only when (build time saved) > (time to move the other ranks' slices over the links):
 *     replicate_ms = (scan + work) nbases
 *     shard_ms     = scan nbases + work nbases / nranks + 3 exchanges' latency + bytes / nranks / link rate
 * with scan = the passes every rank makes over ALL bases (k_part_hist, k_split_bases), work = the passes that shrink with the rank's hash range
 * (both measured on MI355X: 6.0 / 22.3 ps per base - 5.2 ms at 184 Mbp, 58 ms at 2.0 Gbp), bytes = the sparse table words + 8 per distinct k-mer +
 * 8 per offset, every peer's slice on its own xGMI link (direct all-pairs groups).  link_gbs <= 0: NECAT_XGMI_GBS or 100 (of a link's nominal 153).
 * NECAT_INDEX_SHARD=0 / 1 overrides the choice (tests run both); every rank computes the same plan from the same arguments. */
typedef struct {
    int32_t  shard;              /* 1: slices + all-gather is the cheaper plan */
    int32_t  _pad;
    double   replicate_ms, shard_ms, exchange_ms;
    uint64_t exchange_bytes;     /* all ranks' slices together */
} necat_index_plan_t;
int  necat_index_plan(uint64_t nbases, int kmer_size, int nranks, double link_gbs, necat_index_plan_t* out);

/* Test hook: runs the RCCL transport's call path (librccl opened at run time, communicator, send/recv group on the context's
 * stream) with ONE rank sending `bytes` bytes to itself, and compares them. */
int  necat_comm_selftest_rccl(necat_ctx* ctx, uint64_t bytes);
/* The same between TWO devices: rank 0 on the context's device, rank 1 on another one of this process, each sends `bytes` to the other and
 * checks what it received (one ncclSend / ncclRecv group per rank, comm.h's exchange pattern).  Returns 1 (and no error) when the box has fewer
 * than two devices - a test skips then, with that reason. */
int  necat_comm_selftest_rccl2(necat_ctx* ctx, uint64_t bytes);

/* necat_index_build with the work split by hash range and the result all-gathered: the returned index is the
 * complete one, bit-identical to necat_index_build's, on every rank.  Collective: every rank of `comm` calls it.
 * (k < 11 - tables that fit the L2 - are simply built whole on every rank.) */
int  necat_index_build_sharded(necat_ctx* ctx, necat_comm* comm, const necat_volume* ref, int kmer_size, int max_occ,
                               necat_index** out);

/* necat_find_candidates / necat_map_pair for this rank's chunks of the query reads, then the gather-v on `root`.
 * On the root: out = the records of ALL ranks (this rank's first), *n_out their number.  On the other ranks: out =
 * this rank's own records.  *n_local (optional) = this rank's own records, *n_candidates its candidates examined.
 * Collective. */
int  necat_find_candidates_sharded(necat_ctx* ctx, necat_comm* comm, const necat_index* ix, const necat_volume* ref,
                                   const necat_volume* reads, int read_start_id, int ref_start_id, int pairwise,
                                   const necat_map_options* opt, int chunk_reads, int root,
                                   necat_candidate** out, uint64_t* n_out, uint64_t* n_local);
int  necat_map_pair_sharded(necat_ctx* ctx, necat_comm* comm, const necat_index* ix, const necat_volume* ref,
                            const necat_volume* reads, int read_start_id, int ref_start_id, int pairwise,
                            const necat_map_options* opt, int tail_match_len, int chunk_reads, int root,
                            necat_m4** out, uint64_t* n_out, uint64_t* n_local, uint64_t* n_candidates);

/* ---- several volumes on several GPUs (both granularities combined; BASELINE configs[3] / [4]) ---------------------
 * <- the job list of pm_main (pm_worker.c:372-390: reference volume v against query volumes v .. V - 1) and its distribution
 * over grid nodes (necat.pl:190-202), re-cut for the GPUs of one node.  The V (V + 1) / 2 (reference, query) pairs are laid end to
 * end on one cost line (cost ~ bases(query) x bases(reference), halved for the self pair) and rank g takes the g-th of nranks equal
 * stretches; a pair a boundary cuts through is split by query reads - chunks of `chunk_reads` reads, chunk c in slot c % slots, a
 * rank takes a contiguous slot range (necat_amd/csrc/pair_sched.h).  A rank's units are consecutive pairs, so it touches few
 * reference volumes, in ascending order, and the ranks of one reference volume are consecutive ranks: team_lo[v] .. team_hi[v]
 * (inclusive).  A team of several ranks builds that volume's index with necat_index_build_sharded over a communicator of the team.
 * Host-only arithmetic (no device is touched).  Outputs (necat_free each): units = all ranks' units, rank by rank;
 * rank_off[nranks + 1]; team[2 * num_volumes] = (team_lo, team_hi) per reference volume. */
typedef struct { int32_t ref_vol, query_vol, slot_lo, slot_hi; } necat_pair_unit;
/* query reads per chunk for a query volume of `query_reads` reads: 64, fewer for small volumes so that every slot gets several chunks */
int  necat_pair_chunk_reads(uint64_t query_reads, int slots);
int  necat_pair_schedule(const uint64_t* vol_bases, int num_volumes, int nranks, int slots,
                         necat_pair_unit** units, uint64_t** rank_off, int32_t** team);

/* necat_find_candidates / necat_map_pair for ONE share of a pair: only the query chunks c (of chunk_reads reads) with
 * slot_lo <= c % slots < slot_hi are processed, each read exactly as in a whole-pair call - the union of the shares [0, slots) IS the
 * whole pair's record set.  No collective; the records come back on the host as from the whole-pair calls. */
int  necat_find_candidates_part(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                                int read_start_id, int ref_start_id, int pairwise, const necat_map_options* opt,
                                int chunk_reads, int slot_lo, int slot_hi, int slots,
                                necat_candidate** out, uint64_t* n_out);
int  necat_map_pair_part(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                         int read_start_id, int ref_start_id, int pairwise, const necat_map_options* opt, int tail_match_len,
                         int chunk_reads, int slot_lo, int slot_hi, int slots,
                         necat_m4** out, uint64_t* n_out, uint64_t* n_candidates);

/* Test / profiling hook for the dominant kernel: n independent Edlib_align calls
 * (edlib_ex.c:733) on byte-coded (0..3) sequences.  seqs = concatenated fragments, q_off/t_off =
 * start of each fragment in `seqs`.  Outputs per block: edit distance (-1 = fail), qend, tend, and
 * the alignment as one op per column (0 match, 1 ins(query base vs gap), 2 del, 3 mismatch) in
 * forward order, ops_off[i]..ops_off[i+1]. */
int  necat_edlib_align_batch(necat_ctx* ctx, const uint8_t* seqs, uint64_t seqs_len,
                             const uint64_t* q_off, const int32_t* q_len,
                             const uint64_t* t_off, const int32_t* t_len, uint64_t n, double error,
                             int32_t* dist, int32_t* qend, int32_t* tend,
                             uint8_t** ops, uint64_t** ops_off);

int  necat_get_timings(const necat_ctx* ctx, necat_timings* t);
/* the first min(bytes, sizeof(necat_timings)) bytes of the timings: safe for a caller built against an older (smaller) struct */
int  necat_get_timings_sized(const necat_ctx* ctx, void* t, size_t bytes);
void necat_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* NECAT_HIP_H */
