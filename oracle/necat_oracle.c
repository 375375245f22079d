/*
 * necat_oracle.c - CPU restatement of NECAT's overlap hot path.  TEST INFRASTRUCTURE ONLY
 * (see necat_oracle.h).  References are to /root/reference/src/<file>:<line>.
 */
#define _GNU_SOURCE
#include "necat_oracle.h"

#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>

#define ORA_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORA_MAX(a, b) ((a) > (b) ? (a) : (b))

static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "oracle: out of memory (%zu)\n", n); exit(1); } return p; }
static void* xcalloc(size_t n, size_t s) { void* p = calloc(n ? n : 1, s); if (!p) { fprintf(stderr, "oracle: out of memory (%zu x %zu)\n", n, s); exit(1); } return p; }
static void* xrealloc(void* q, size_t n) { void* p = realloc(q, n ? n : 1); if (!p) { fprintf(stderr, "oracle: out of memory (%zu)\n", n); exit(1); } return p; }
static double now_sec(void) { struct timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec + 1e-6 * tv.tv_usec; }

/* ===================================================================== volumes */

static const char kPacMagic[] = "ontcns_pac_header_hofuwhogfuewo"; /* packed_db.c:7 */

static inline uint8_t pac_get(const uint8_t* pac, uint64_t l) /* ontcns_aux.h:119 */
{
    return (pac[l >> 2] >> ((~l & 3) << 1)) & 3;
}

int ora_volume_load(const char* path, ora_volume* v) /* packed_db.c:317-345 */
{
    memset(v, 0, sizeof(*v));
    FILE* in = fopen(path, "rb");
    if (!in) return -1;
    char magic[64];
    size_t ml = strlen(kPacMagic);
    if (fread(magic, 1, ml, in) != ml || memcmp(magic, kPacMagic, ml)) { fclose(in); return -2; }
    uint64_t ns, nb;
    if (fread(&ns, 8, 1, in) != 1 || fread(&nb, 8, 1, in) != 1) { fclose(in); return -3; }
    v->nseq = ns; v->nbases = nb;
    v->offset = xmalloc(8 * ns); v->size = xmalloc(8 * ns); v->hdr_offset = xmalloc(8 * ns);
    for (uint64_t i = 0; i < ns; ++i) {
        uint64_t rec[4];
        if (fread(rec, 32, 1, in) != 1) { fclose(in); return -3; }
        v->offset[i] = rec[0]; v->size[i] = rec[1]; v->hdr_offset[i] = rec[2];
    }
    if (fread(&v->hdr_bytes, 8, 1, in) != 1) { fclose(in); return -3; }
    v->hdr = xmalloc(v->hdr_bytes + 1);
    if (v->hdr_bytes && fread(v->hdr, 1, v->hdr_bytes, in) != v->hdr_bytes) { fclose(in); return -3; }
    v->hdr[v->hdr_bytes] = 0;
    uint64_t pb = (nb + 3) >> 2;
    v->pac = xmalloc(pb + 8);
    if (pb && fread(v->pac, 1, pb, in) != pb) { fclose(in); return -3; }
    fclose(in);
    return 0;
}

void ora_volume_free(ora_volume* v)
{
    free(v->pac); free(v->offset); free(v->size); free(v->hdr_offset); free(v->hdr);
    memset(v, 0, sizeof(*v));
}

void ora_volume_extract(const ora_volume* v, uint64_t i, int strand, uint8_t* out) /* packed_db.c:255-277 */
{
    uint64_t s = v->offset[i], e = s + v->size[i], pos = 0;
    if (strand == 0) {
        for (uint64_t k = s; k < e; ++k) out[pos++] = pac_get(v->pac, k);
    } else {
        for (uint64_t k = e; k != s; --k) out[pos++] = 3 - pac_get(v->pac, k - 1);
    }
}

uint64_t ora_offset_to_id(const ora_volume* v, uint64_t offset) /* packed_db.c:173-189 */
{
    uint64_t ns = v->nseq, left = 0, mid = 0, right = ns;
    while (left < right) {
        mid = (left + right) >> 1;
        if (offset >= v->offset[mid]) {
            if (mid == ns - 1) break;
            if (offset < v->offset[mid + 1]) break;
            left = mid + 1;
        } else {
            right = mid;
        }
    }
    return mid;
}

int ora_volumes_info_load(const char* wrk_dir, ora_volumes_info* vi) /* makedb_aux.c:36-118 */
{
    memset(vi, 0, sizeof(*vi));
    char path[4096];
    size_t n = strlen(wrk_dir);
    const char* sep = (n && wrk_dir[n - 1] == '/') ? "" : "/";
    snprintf(path, sizeof path, "%s%sreads_info.txt", wrk_dir, sep);
    FILE* in = fopen(path, "r");
    if (!in) return -1;
    if (fscanf(in, "%d%d", &vi->num_volumes, &vi->num_reads) != 2) { fclose(in); return -2; }
    fclose(in);
    snprintf(path, sizeof path, "%s%svolume_names.txt", wrk_dir, sep);
    in = fopen(path, "r");
    if (!in) return -1;
    vi->names = xcalloc(vi->num_volumes, sizeof(char*));
    vi->read_start_id = xcalloc(vi->num_volumes, sizeof(int));
    vi->read_count = xcalloc(vi->num_volumes, sizeof(int));
    char line[4096];
    for (int i = 0; i < vi->num_volumes; ++i) {
        if (!fgets(line, sizeof line, in)) { fclose(in); return -2; }
        size_t k = 0, L = strlen(line);
        while (k < L && !isspace((unsigned char)line[k])) ++k;
        vi->names[i] = xmalloc(k + 1);
        memcpy(vi->names[i], line, k); vi->names[i][k] = 0;
        ++k;
        vi->read_start_id[i] = atoi(line + k);
        while (k < L && !isspace((unsigned char)line[k])) ++k;
        ++k;
        vi->read_count[i] = atoi(line + k);
    }
    fclose(in);
    return 0;
}

void ora_volumes_info_free(ora_volumes_info* vi)
{
    if (vi->names) for (int i = 0; i < vi->num_volumes; ++i) free(vi->names[i]);
    free(vi->names); free(vi->read_start_id); free(vi->read_count);
    memset(vi, 0, sizeof(*vi));
}

/* ===================================================================== options */

void ora_options_default(ora_options* o) /* map_options.c:12-28 */
{
    o->kmer_size = 15; o->scan_window = 10; o->kmer_cnt_cutoff = 500; o->block_size = 2000;
    o->block_score_cutoff = 3; o->num_candidates = 500; o->align_size_cutoff = 500;
    o->ddfs_cutoff = 0.25; o->error = 0.5; o->num_output = 500; o->num_threads = 1;
    o->job = 1; o->binary_output = 0; o->use_hdr_as_id = 1;
}

int ora_options_parse(int argc, char** argv, ora_options* o) /* map_options.c:90-150, flags :10 */
{
    optind = 1;
    int c;
    while ((c = getopt(argc, argv, "k:z:q:b:s:n:a:d:e:m:t:j:u:i:")) != -1) {
        switch (c) {
        case 'k': o->kmer_size = atoi(optarg); break;
        case 'z': o->scan_window = atoi(optarg); break;
        case 'q': o->kmer_cnt_cutoff = atoi(optarg); break;
        case 'b': o->block_size = atoi(optarg); break;
        case 's': o->block_score_cutoff = atoi(optarg); break;
        case 'n': o->num_candidates = atoi(optarg); break;
        case 'a': o->align_size_cutoff = atoi(optarg); break;
        case 'd': o->ddfs_cutoff = atof(optarg); break;
        case 'e': o->error = atof(optarg); break;
        case 'm': o->num_output = atoi(optarg); break;
        case 't': o->num_threads = atoi(optarg); break;
        case 'j': o->job = atoi(optarg); break;
        case 'u': o->binary_output = atoi(optarg); break;
        case 'i': o->use_hdr_as_id = atoi(optarg); break;
        default: return -1;
        }
    }
    return 0;
}

/* ===================================================================== k-mer index */

#define ORA_OFFSET_BITS 34
#define ORA_OFFSET_MASK ((1ULL << ORA_OFFSET_BITS) - 1)

/*
 * lookup_table.c:15-58 (get_kmer_counts), :60-92 (get_offset_list), hash_list_bucket_sort.c:134
 * (stable LSD radix sort on the hash), :94-126 (build_kmer_starts, clear_hash_in_offset_list).
 * Net effect restated: count every k-mer occurrence (k-mers never span reads, earlier base is more
 * significant), drop k-mers occurring more than max_occ times, list the base offsets of the
 * surviving occurrences grouped by hash in ascending hash order and, inside a hash, in ascending
 * offset order (scan order + stable sort).  A counting sort by hash gives exactly that order.
 */
ora_index* ora_index_build(const ora_volume* ref, int k, int max_occ)
{
    ora_index* ix = xcalloc(1, sizeof(*ix));
    ix->k = k;
    uint64_t T = 1ULL << (2 * k), mask = T - 1;
    uint32_t* cnt = xcalloc(T, sizeof(uint32_t));
    for (uint64_t i = 0; i < ref->nseq; ++i) {
        uint64_t off = ref->offset[i], sz = ref->size[i], h = 0;
        for (uint64_t j = 0; j < sz; ++j) {
            h = ((h << 2) | pac_get(ref->pac, off + j)) & mask;
            if (j + 1 >= (uint64_t)k) { if (cnt[h] != UINT32_MAX) ++cnt[h]; }
        }
    }
    ix->kmer_stats = xmalloc(T * sizeof(uint64_t));
    uint64_t n = 0;
    for (uint64_t h = 0; h < T; ++h) {
        uint64_t c = cnt[h];
        if (c > (uint64_t)max_occ) c = 0;            /* lookup_table.c:44 */
        ix->kmer_stats[h] = (c << ORA_OFFSET_BITS) | (c ? n : 0); /* start index; 0 when absent */
        n += c;
    }
    free(cnt);
    ix->n_offsets = n;
    ix->offset_list = xmalloc(n * sizeof(uint64_t));
    /* fill in scan order; low bits of kmer_stats act as the write cursor, restored afterwards */
    for (uint64_t i = 0; i < ref->nseq; ++i) {
        uint64_t off = ref->offset[i], sz = ref->size[i], h = 0;
        for (uint64_t j = 0; j < sz; ++j) {
            h = ((h << 2) | pac_get(ref->pac, off + j)) & mask;
            if (j + 1 >= (uint64_t)k) {
                uint64_t u = ix->kmer_stats[h];
                if (u >> ORA_OFFSET_BITS) {
                    ix->offset_list[u & ORA_OFFSET_MASK] = off + j + 1 - k;
                    ix->kmer_stats[h] = u + 1;
                }
            }
        }
    }
    for (uint64_t h = 0; h < T; ++h) {
        uint64_t u = ix->kmer_stats[h], c = u >> ORA_OFFSET_BITS;
        if (c) ix->kmer_stats[h] = u - c;
    }
    return ix;
}

void ora_index_free(ora_index* ix)
{
    if (!ix) return;
    free(ix->kmer_stats); free(ix->offset_list); free(ix);
}

/* ===================================================================== seeding */

#define BLK_SEEDS 40   /* word_finder_aux.h:9 */
#define DDFS_CUTOFF 0.25 /* word_finder.c:10 (the -d flag is parsed but ignored) */

typedef struct {       /* word_finder_aux.h:19-25 */
    short score;
    short blk_offset[BLK_SEEDS];
    int   kmer_id[BLK_SEEDS];
    int   last_kmer_id;
    int   index;
} ScoringBlock;
typedef struct { int score; int block_idx; } ScoringBlockIndex; /* :27-30 */
typedef struct { uint64_t qoff, soff; } ChainSeed;              /* :11-14 */

struct ora_wfd {
    ScoringBlock* _blk; ScoringBlock* blk; ScoringBlockIndex* idx; int nblk; int nalloc;
    uint64_t* hash; size_t nhash, mhash;
    ChainSeed* cs; size_t ncs, mcs;
    /* chain dp scratch (chain_dp.h:8-23) */
    int *f, *p, *t, *v; size_t mdp;
    int (*u)[2]; size_t mu;
    ora_can_vec lcan;
    int kmer_size, max_dist, bw, max_skip, min_cnt, min_sc;
};

static void can_push(ora_can_vec* v, const ora_candidate* c)
{
    if (v->n == v->m) { v->m = v->m ? v->m * 2 : 16; v->a = xrealloc(v->a, v->m * sizeof(*v->a)); }
    v->a[v->n++] = *c;
}

ora_wfd* ora_wfd_new(uint64_t reference_bases, int block_size, int kmer_size, int block_score_cutoff)
{   /* word_finder.c:15-38, chain_dp.c:161-181 */
    ora_wfd* w = xcalloc(1, sizeof(*w));
    int nblk = (int)(reference_bases / (uint64_t)block_size + 5);
    w->nalloc = nblk;
    w->_blk = xmalloc(sizeof(ScoringBlock) * ((size_t)nblk + 1));
    w->blk = w->_blk + 1;
    w->idx = xmalloc(sizeof(ScoringBlockIndex) * (size_t)nblk);
    for (int i = 0; i < nblk; ++i) { w->blk[i].index = -1; w->blk[i].score = 0; w->blk[i].last_kmer_id = -1; }
    w->_blk[0].score = 0;
    w->nblk = 0;
    w->kmer_size = kmer_size; w->max_dist = 5000; w->bw = 500; w->max_skip = 25;
    w->min_cnt = block_score_cutoff; w->min_sc = 30;
    return w;
}

void ora_wfd_free(ora_wfd* w)
{
    if (!w) return;
    free(w->_blk); free(w->idx); free(w->hash); free(w->cs);
    free(w->f); free(w->p); free(w->t); free(w->v); free(w->u); free(w->lcan.a); free(w);
}

static void wfd_clear(ora_wfd* w) /* word_finder.c:40-52 */
{
    for (int i = 0; i < w->nblk; ++i) {
        int b = w->idx[i].block_idx;
        w->blk[b].index = -1; w->blk[b].score = 0; w->blk[b].last_kmer_id = -1;
    }
    w->nhash = 0; w->ncs = 0; w->nblk = 0;
}

static void fill_one_seed(ora_wfd* w, int kmer_id, int blk_id, short blk_offset) /* word_finder.c:85-104 */
{
    ScoringBlock* sb = w->blk + blk_id;
    if (sb->last_kmer_id >= kmer_id + 1) return;
    if (sb->score >= BLK_SEEDS) return;
    int sid = sb->score;
    ++sb->score;
    sb->blk_offset[sid] = blk_offset;
    sb->kmer_id[sid] = kmer_id + 1;
    sb->last_kmer_id = kmer_id + 1;
    if (sb->index == -1) {
        sb->index = w->nblk++;
        w->idx[sb->index].block_idx = blk_id;
    }
    w->idx[sb->index].score = sb->score + (sb - 1)->score;
}

static void collect_seeds(ora_wfd* w, const uint8_t* read, int read_size, int read_id, int read_start_id,
                          int reference_start_id, const ora_volume* ref, const ora_index* ix,
                          int block_size, int kmer_size, int scan_window, int pairwise)
{   /* word_finder.c:107-139 + extract_hash_values :66-83 */
    uint64_t soff_max = UINT64_MAX;
    if (pairwise) {
        int max_rid = reference_start_id + (int)ref->nseq;
        if (read_id + read_start_id >= reference_start_id && read_id + read_start_id < max_rid)
            soff_max = ref->offset[read_id];
    }
    w->nhash = 0;
    for (int i = 0; i <= read_size - kmer_size; i += scan_window) {
        uint64_t h = 0;
        for (int j = 0; j < kmer_size; ++j) h = (h << 2) | read[i + j];
        if (w->nhash == w->mhash) { w->mhash = w->mhash ? w->mhash * 2 : 1024; w->hash = xrealloc(w->hash, 8 * w->mhash); }
        w->hash[w->nhash++] = h;
    }
    for (int i = 0; i < (int)w->nhash; ++i) {
        uint64_t u = ix->kmer_stats[w->hash[i]];        /* lookup_table.c:176-190 */
        uint64_t cnt = u >> ORA_OFFSET_BITS, start = u & ORA_OFFSET_MASK;
        const uint64_t* list = ix->offset_list + start;
        for (uint64_t k = 0; k < cnt; ++k) {
            if (list[k] >= soff_max) continue;
            fill_one_seed(w, i, (int)(list[k] / (uint64_t)block_size), (short)(list[k] % (uint64_t)block_size));
        }
    }
}

/* word_finder.c:141-168.  NB: quotient in float, "- 1.0" and compare in double. */
static inline int ddf_ok(int dloc, int dseed, float scan_window)
{
    return fabs(dloc / (dseed * scan_window) - 1.0) < DDFS_CUTOFF;
}

static int scoring_seeds(const int* t_loc, const int* t_seedn, int* t_score, int* loc, int k, int* rep_loc,
                         float scan_window, int read_size)
{
    int i, j, maxval = 0, maxi = 0, rep = 0, lasti = 0, tempi;
    for (i = 0; i < k; i++) t_score[i] = 0;
    for (i = 0; i < k - 1; i++)
        for (j = i + 1, tempi = t_seedn[i]; j < k; j++)
            if (tempi != t_seedn[j] && t_seedn[j] - t_seedn[i] > 0 && t_loc[j] - t_loc[i] > 0 &&
                t_loc[j] - t_loc[i] < read_size && ddf_ok(t_loc[j] - t_loc[i], t_seedn[j] - t_seedn[i], scan_window)) {
                t_score[i]++; t_score[j]++; tempi = t_seedn[j];
            }
    for (i = 0; i < k; i++) {
        if (maxval < t_score[i]) { maxval = t_score[i]; maxi = i; rep = 0; }
        else if (maxval == t_score[i]) { rep++; lasti = i; }
    }
    for (i = 0; i < 4; i++) loc[i] = 0;
    if (maxval >= 5 && rep == maxval) {
        loc[0] = t_loc[maxi]; loc[1] = t_seedn[maxi]; *rep_loc = maxi; loc[2] = t_loc[lasti]; loc[3] = t_seedn[lasti];
        return 1;
    } else if (maxval >= 5 && rep != maxval) {
        for (j = 0; j < maxi; j++)
            if (t_seedn[maxi] - t_seedn[j] > 0 && t_loc[maxi] - t_loc[j] > 0 && t_loc[maxi] - t_loc[j] < read_size &&
                ddf_ok(t_loc[maxi] - t_loc[j], t_seedn[maxi] - t_seedn[j], scan_window)) {
                if (loc[0] == 0) { loc[0] = t_loc[j]; loc[1] = t_seedn[j]; *rep_loc = j; }
                else { loc[2] = t_loc[j]; loc[3] = t_seedn[j]; }
            }
        j = maxi;
        if (loc[0] == 0) { loc[0] = t_loc[j]; loc[1] = t_seedn[j]; *rep_loc = j; }
        else { loc[2] = t_loc[j]; loc[3] = t_seedn[j]; }
        for (j = maxi + 1; j < k; j++)
            if (t_seedn[j] - t_seedn[maxi] > 0 && t_loc[j] - t_loc[maxi] > 0 && t_loc[j] - t_loc[maxi] <= read_size &&
                ddf_ok(t_loc[j] - t_loc[maxi], t_seedn[j] - t_seedn[maxi], scan_window)) {
                if (loc[0] == 0) { loc[0] = t_loc[j]; loc[1] = t_seedn[j]; *rep_loc = j; }
                else { loc[2] = t_loc[j]; loc[3] = t_seedn[j]; }
            }
        return 1;
    }
    return 0;
}

/* ---- sorting helpers: every comparator on this path is a strict total order on the data it sees
 * (SURVEY.md §8a a16), so any correct sort reproduces klib's introsort result. ---- */
static int cmp_chain_seed(const void* a, const void* b) /* word_finder_aux.h:17 ChainSeedLT */
{
    const ChainSeed* x = a; const ChainSeed* y = b;
    if (x->soff != y->soff) return x->soff < y->soff ? -1 : 1;
    if (x->qoff != y->qoff) return x->qoff < y->qoff ? -1 : 1;
    return 0;
}
static int cmp_intpair_gt(const void* a, const void* b) /* chain_dp.c:8 */
{
    const int* x = a; const int* y = b;
    if (x[0] != y[0]) return x[0] > y[0] ? -1 : 1;
    if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
    return 0;
}
static int cmp_can_cdp(const void* a, const void* b) /* chain_dp.c:26-32 */
{
    const ora_candidate* x = a; const ora_candidate* y = b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    if (x->qoff != y->qoff) return x->qoff < y->qoff ? -1 : 1;
    if (x->soff != y->soff) return x->soff < y->soff ? -1 : 1;
    return 0;
}
static int cmp_can_pm(const void* a, const void* b) /* pm_worker.c:16-24 */
{
    const ora_candidate* x = a; const ora_candidate* y = b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    if (x->qdir != y->qdir) return x->qdir < y->qdir ? -1 : 1;
    if (x->sid != y->sid) return x->sid < y->sid ? -1 : 1;
    if (x->qoff != y->qoff) return x->qoff < y->qoff ? -1 : 1;
    if (x->soff != y->soff) return x->soff < y->soff ? -1 : 1;
    return 0;
}

static inline int ilog2_32(uint32_t v) /* chain_dp.c:18-23 */
{
    int r = -1;
    while (v) { ++r; v >>= 1; }
    return r;
}

/* chain_dp.c:37-159 */
static void chain_dp(ora_wfd* w, int qid, int qdir, uint64_t qsize, int sid, uint64_t ssize)
{
    const ChainSeed* seeds = w->cs;
    const int n_seeds = (int)w->ncs;
    const int kmer_size = w->kmer_size, max_dist = w->max_dist, bw = w->bw, max_skip = w->max_skip;
    const int min_cnt = w->min_cnt, min_sc = w->min_sc;
    w->lcan.n = 0;
    if ((size_t)n_seeds > w->mdp) {
        w->mdp = (size_t)n_seeds * 2;
        w->f = xrealloc(w->f, 4 * w->mdp); w->p = xrealloc(w->p, 4 * w->mdp);
        w->t = xrealloc(w->t, 4 * w->mdp); w->v = xrealloc(w->v, 4 * w->mdp);
    }
    int *f = w->f, *p = w->p, *t = w->t, *v = w->v;
    for (int i = 0; i < n_seeds; ++i) { f[i] = 0; p[i] = -1; t[i] = 0; v[i] = 0; }
    int i, j, k, st = 0;
    for (i = 0; i < n_seeds; ++i) {
        uint64_t ri = seeds[i].soff, qi = seeds[i].qoff;
        int max_j = -1, max_f = kmer_size, n_skip = 0;
        while (st < i && ri - seeds[st].soff > (uint64_t)max_dist) ++st;
        for (j = i - 1; j >= st; --j) {
            if (ri <= seeds[j].soff || qi <= seeds[j].qoff || qi - seeds[j].qoff > (uint64_t)max_dist) continue;
            uint64_t dr = ri - seeds[j].soff, dq = qi - seeds[j].qoff;
            uint64_t dd = dr > dq ? dr - dq : dq - dr;
            if (dd > (uint64_t)bw) continue;
            uint64_t min_d = ORA_MIN(dq, dr);
            int sc = (int)ORA_MIN(min_d, (uint64_t)kmer_size);
            int log_dd = dd ? ilog2_32((uint32_t)dd) : 0;
            sc -= (int)(dd * 0.01 * kmer_size) + (log_dd >> 1);
            sc += f[j];
            if (sc > max_f) {
                max_f = sc; max_j = j;
                if (n_skip > 0) --n_skip;
            } else if (t[j] == i) {
                if (++n_skip > max_skip) break;
            }
            if (p[j] >= 0) t[p[j]] = i;
        }
        f[i] = max_f; p[i] = max_j;
        v[i] = (max_j >= 0 && v[max_j] > max_f) ? v[max_j] : max_f;
    }
    memset(t, 0, sizeof(int) * (size_t)n_seeds);
    for (i = 0; i < n_seeds; ++i) if (p[i] >= 0) t[p[i]] = 1;
    int n_u = 0;
    for (i = 0; i < n_seeds; ++i) if (t[i] == 0 && v[i] >= min_sc) ++n_u;
    if (n_u == 0) return;
    if ((size_t)n_u > w->mu) { w->mu = (size_t)n_u * 2; w->u = xrealloc(w->u, sizeof(int[2]) * w->mu); }
    for (i = n_u = 0; i < n_seeds; ++i) {
        if (t[i] == 0 && v[i] >= min_sc) {
            j = i;
            while (j >= 0 && f[j] < v[j]) j = p[j];
            if (j < 0) j = i;
            w->u[n_u][0] = f[j]; w->u[n_u][1] = j; ++n_u;
        }
    }
    qsort(w->u, (size_t)n_u, sizeof(int[2]), cmp_intpair_gt);

    ora_candidate can;
    memset(&can, 0, sizeof can);
    can.qid = qid; can.qdir = qdir; can.qsize = (int64_t)qsize; can.sid = sid; can.sdir = 0; can.ssize = (int64_t)ssize;
    memset(t, 0, sizeof(int) * (size_t)n_seeds);
    int n_v = 0;
    for (i = n_v = k = 0; i < n_u; ++i) {
        int n_v0 = n_v, k0 = k;
        j = w->u[i][1];
        can.qend = (int64_t)seeds[j].qoff + kmer_size;
        can.send = (int64_t)seeds[j].soff + kmer_size;
        can.qoff = can.qend; can.soff = can.send;
        int last_j = j;
        do { last_j = j; n_v++; t[j] = 1; j = p[j]; } while (j >= 0 && t[j] == 0);
        if (j < 0) {
            if (n_v - n_v0 >= min_cnt) {
                can.qbeg = (int64_t)seeds[last_j].qoff; can.sbeg = (int64_t)seeds[last_j].soff;
                can.score = w->u[i][0];
                can_push(&w->lcan, &can); ++k;
            }
        } else if (w->u[i][0] - f[j] >= min_sc) {
            if (n_v - n_v0 >= min_cnt) {
                can.qbeg = (int64_t)seeds[last_j].qoff; can.sbeg = (int64_t)seeds[last_j].soff;
                can.score = w->u[i][0] - f[j];
                can_push(&w->lcan, &can); ++k;
            }
        }
        if (k0 == k) n_v = n_v0;
    }
    if (w->lcan.n == 0) return;
    qsort(w->lcan.a, w->lcan.n, sizeof(ora_candidate), cmp_can_cdp);
}

static void cs_push(ora_wfd* w, uint64_t qoff, uint64_t soff)
{
    if (w->ncs == w->mcs) { w->mcs = w->mcs ? w->mcs * 2 : 256; w->cs = xrealloc(w->cs, sizeof(ChainSeed) * w->mcs); }
    w->cs[w->ncs].qoff = qoff; w->cs[w->ncs].soff = soff; ++w->ncs;
}

static void clear_block_scores(ora_wfd* w, const ora_candidate* can, uint64_t subject_start, uint64_t block_size)
{   /* word_finder.c:171-182 */
    uint64_t sblk = ((uint64_t)can->sbeg + subject_start) / block_size;
    uint64_t eblk = ((uint64_t)can->send + subject_start) / block_size;
    for (uint64_t i = sblk; i <= eblk; ++i) w->blk[i].score = 0;
}

/* word_finder.c:184-360 */
static int find_candidate_for_one_block(ora_wfd* w, int block_id, const ora_volume* ref, int block_score_cutoff,
                                        uint64_t block_size, int align_size_cutoff, int scan_window,
                                        int qid, int qdir, uint64_t qsize, ora_can_vec* out)
{
    int kmer_id_list[BLK_SEEDS * 2], blk_offset_list[BLK_SEEDS * 2], score_list[BLK_SEEDS * 2];
    int n_seeds = 0, A = 0;
    uint64_t blk_start = block_size * (uint64_t)block_id;
    if (w->blk[block_id - 1].score) {
        ScoringBlock* sb = w->blk + block_id - 1;
        for (int i = 0; i < sb->score; ++i) { kmer_id_list[n_seeds] = sb->kmer_id[i]; blk_offset_list[n_seeds] = sb->blk_offset[i]; ++n_seeds; }
        A = (int)block_size;
        blk_start = block_size * (uint64_t)(block_id - 1);
    }
    {
        ScoringBlock* sb = w->blk + block_id;
        for (int i = 0; i < sb->score; ++i) { kmer_id_list[n_seeds] = sb->kmer_id[i]; blk_offset_list[n_seeds] = sb->blk_offset[i] + A; ++n_seeds; }
    }
    int max_score_id = -1, sc4[4];
    int r = scoring_seeds(blk_offset_list, kmer_id_list, score_list, sc4, n_seeds, &max_score_id, (float)scan_window, (int)qsize);
    if (!r) return 0;
    if (score_list[max_score_id] < block_score_cutoff) return 0;

    uint64_t seed_toff = (uint64_t)sc4[0] + blk_start;
    uint64_t seed_qoff = (uint64_t)(int64_t)((sc4[1] - 1) * scan_window);
    int seed_bid = (int)(seed_toff / block_size);
    uint64_t seed_tid = ora_offset_to_id(ref, seed_toff);
    uint64_t seed_tsize = ref->size[seed_tid], seed_tstart = ref->offset[seed_tid];
    uint64_t seed_tend = seed_tstart + seed_tsize;
    seed_toff -= seed_tstart;
    uint64_t L = ORA_MIN(seed_toff, seed_qoff);
    int bid_start = (int)((uint64_t)seed_bid - L / block_size - 1);
    if (bid_start < 0) bid_start = 0;
    uint64_t tr = seed_tsize - seed_toff, qr = qsize - seed_qoff;
    L = ORA_MIN(tr, qr);
    int bid_end = (int)((uint64_t)seed_bid + (L + block_size - 1) / block_size);
    w->ncs = 0;

    int seed_score = 0;
    for (int i = bid_start; i <= seed_bid; ++i) {
        ScoringBlock* sb = w->blk + i;
        if (!sb->score) continue;
        blk_start = (uint64_t)i * block_size;
        int relevant = 0;
        for (int k = 0; k < sb->score; ++k) {
            uint64_t toff = blk_start + (uint64_t)(int64_t)sb->blk_offset[k];
            uint64_t qoff = (uint64_t)(int64_t)((sb->kmer_id[k] - 1) * scan_window);
            if (toff < seed_tstart) continue;
            toff -= seed_tstart;
            if (toff < seed_toff && qoff < seed_qoff) {
                double s = 1.0 * (seed_toff - toff) / (seed_qoff - qoff) - 1.0;
                if (!(fabs(s) < DDFS_CUTOFF)) continue;
                ++relevant;
                cs_push(w, qoff, toff);
            }
        }
        if (i != seed_bid && 1.0 * relevant / sb->score >= 0.4) sb->score = 0;
        seed_score += relevant;
    }
    cs_push(w, seed_qoff, seed_toff);
    for (int i = seed_bid; i <= bid_end; ++i) {
        ScoringBlock* sb = w->blk + i;
        if (!sb->score) continue;
        blk_start = (uint64_t)i * block_size;
        int relevant = 0;
        for (int k = 0; k < sb->score; ++k) {
            uint64_t toff = blk_start + (uint64_t)(int64_t)sb->blk_offset[k];
            uint64_t qoff = (uint64_t)(int64_t)((sb->kmer_id[k] - 1) * scan_window);
            if (toff >= seed_tend) continue;
            toff -= seed_tstart;
            if (toff > seed_toff && qoff > seed_qoff) {
                double s = 1.0 * (toff - seed_toff) / (qoff - seed_qoff) - 1.0;
                if (!(fabs(s) < DDFS_CUTOFF)) continue;
                ++relevant;
                cs_push(w, qoff, toff);
            }
        }
        if (i != seed_bid && 1.0 * relevant / sb->score >= 0.4) sb->score = 0;
        seed_score += relevant;
    }

    qsort(w->cs, w->ncs, sizeof(ChainSeed), cmp_chain_seed);
    chain_dp(w, qid, qdir, qsize, (int)seed_tid, seed_tsize);
    size_t ncan = w->lcan.n;
    if (!ncan) return 0;

#define CONTAINS_ANCHOR(c) ((int64_t)seed_qoff >= (c).qbeg && (int64_t)seed_qoff < (c).qend && \
                            (int64_t)seed_toff >= (c).sbeg && (int64_t)seed_toff < (c).send)
#define EMIT(c) do { \
        (c).score = seed_score; (c).qoff = (int64_t)seed_qoff; (c).soff = (int64_t)seed_toff; \
        clear_block_scores(w, &(c), seed_tstart, block_size); \
        int ok = ((c).send - (c).sbeg >= align_size_cutoff) || ((c).qend - (c).qbeg >= align_size_cutoff); \
        if (ok) can_push(out, &(c)); \
        return ok; } while (0)

    ora_candidate can = w->lcan.a[0];
    if (CONTAINS_ANCHOR(can)) EMIT(can);
    size_t max_i = ncan;
    int max_cov = 0;
    for (size_t i = 0; i < ncan; ++i) {
        can = w->lcan.a[i];
        if (CONTAINS_ANCHOR(can)) {
            int cov = (int)(can.qend - can.qbeg);
            if (cov > max_cov) { max_cov = cov; max_i = i; }
        }
    }
    /* word_finder.c:335-343: NB the emitted record is `can` as left by the loop, i.e. the LAST
     * chain of the list, not lcanv[max_i] (reference behaviour, kept). */
    if (max_i < ncan) EMIT(can);
    can = w->lcan.a[0];
    if (can.qend - can.qbeg >= 5000) EMIT(can);
    return 0;
#undef EMIT
#undef CONTAINS_ANCHOR
}

void ora_find_candidates(const uint8_t* read, int read_size, int qid, int qdir,
                         int read_start_id, int reference_start_id, int pairwise,
                         const ora_volume* reference, const ora_index* ix,
                         const ora_options* opt, ora_wfd* w, ora_can_vec* out)
{   /* word_finder.c:364-412 */
    wfd_clear(w);
    collect_seeds(w, read, read_size, qid, read_start_id, reference_start_id, reference, ix,
                  opt->block_size, opt->kmer_size, opt->scan_window, pairwise);
    for (int i = 0; i < w->nblk; ++i) {
        int b = w->idx[i].block_idx;
        if (w->blk[b].score >= opt->block_score_cutoff)
            if (w->idx[i].score >= 2 * opt->block_score_cutoff)
                find_candidate_for_one_block(w, b, reference, 2 * opt->block_score_cutoff,
                                             (uint64_t)opt->block_size, opt->align_size_cutoff,
                                             opt->scan_window, qid, qdir, (uint64_t)read_size, out);
    }
}

/* ===================================================================== block Myers / edlib */

typedef uint64_t Word;
#define WORD_SIZE 64
#define HIGH_BIT (1ULL << 63)
#define ORA_MAXW 64       /* edlib_ex_aux.h:20-21 MaxNumBlocks */
#define ORA_MAXSEQ 4096   /* MaxSeqSize */

typedef struct { char* s; size_t n, m; } ostr;
static void ostr_clear(ostr* o) { o->n = 0; if (o->s) o->s[0] = 0; }
static void ostr_append(ostr* o, const char* src, size_t len)
{
    if (o->n + len + 1 > o->m) { o->m = (o->n + len + 1) * 2; o->s = xrealloc(o->s, o->m); }
    memcpy(o->s + o->n, src, len); o->n += len; o->s[o->n] = 0;
}
static void ostr_putc(ostr* o, char c) { ostr_append(o, &c, 1); }

struct ora_aligner {
    double error;
    Word* peq;                 /* [5][ORA_MAXW] */
    Word *bP, *bM; int* bS;    /* running column state */
    Word *Ps, *Ms; int* Sc;    /* [cols][ORA_MAXW] */
    int *first, *last;         /* [cols] */
    int* ends; size_t nends, mends;
    unsigned char* ops; size_t nops, mops;
    /* onc_align buffers */
    char *qabuf, *tabuf;
    uint8_t *qfrag, *tfrag;
    ostr rq, rt, fq, ft, qa, ta;
};

ora_aligner* ora_aligner_new(double error)
{
    ora_aligner* a = xcalloc(1, sizeof(*a));
    a->error = error;
    a->peq = xmalloc(sizeof(Word) * 5 * ORA_MAXW);
    a->bP = xmalloc(sizeof(Word) * ORA_MAXW); a->bM = xmalloc(sizeof(Word) * ORA_MAXW); a->bS = xmalloc(sizeof(int) * ORA_MAXW);
    a->Ps = xmalloc(sizeof(Word) * ORA_MAXW * ORA_MAXSEQ); a->Ms = xmalloc(sizeof(Word) * ORA_MAXW * ORA_MAXSEQ);
    a->Sc = xmalloc(sizeof(int) * ORA_MAXW * ORA_MAXSEQ);
    a->first = xmalloc(sizeof(int) * ORA_MAXSEQ); a->last = xmalloc(sizeof(int) * ORA_MAXSEQ);
    a->qabuf = xmalloc(100000); a->tabuf = xmalloc(100000); /* oc_aligner.c:24-25 */
    a->qfrag = xmalloc(ORA_MAXSEQ); a->tfrag = xmalloc(ORA_MAXSEQ);
    return a;
}

void ora_aligner_free(ora_aligner* a)
{
    if (!a) return;
    free(a->peq); free(a->bP); free(a->bM); free(a->bS); free(a->Ps); free(a->Ms); free(a->Sc);
    free(a->first); free(a->last); free(a->ends); free(a->ops); free(a->qabuf); free(a->tabuf);
    free(a->qfrag); free(a->tfrag); free(a->rq.s); free(a->rt.s); free(a->fq.s); free(a->ft.s); free(a->qa.s); free(a->ta.s);
    free(a);
}

static inline int nwords(int n) { return (n + WORD_SIZE - 1) / WORD_SIZE; }

static void build_peq(const uint8_t* q, int qn, Word* peq) /* edlib_ex.c:36-54 */
{
    int nb = nwords(qn);
    for (int s = 0; s <= 4; ++s)
        for (int b = 0; b < nb; ++b) {
            Word w = 0;
            if (s < 4) {
                for (int r = (b + 1) * WORD_SIZE - 1; r >= b * WORD_SIZE; --r) {
                    w <<= 1;
                    if (r >= qn || q[r] == s) w += 1;
                }
            } else w = (Word)-1;
            peq[s * ORA_MAXW + b] = w;
        }
}

/* edlib_ex.c:71-106 calculateBlock (Myers' Advance_Block) */
static inline int advance_block(Word Pv, Word Mv, Word Eq, int hin, Word* PvOut, Word* MvOut)
{
    Word hinIsNeg = (Word)(hin >> 2) & 1ULL;
    Word Xv = Eq | Mv;
    Eq |= hinIsNeg;
    Word Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    Word Ph = Mv | ~(Xh | Pv);
    Word Mh = Pv & Xh;
    int hout = (int)((Ph & HIGH_BIT) >> 63);
    hout -= (int)((Mh & HIGH_BIT) >> 63);
    Ph <<= 1; Mh <<= 1;
    Mh |= hinIsNeg;
    Ph |= (Word)((hin + 1) >> 1);
    *PvOut = Mh | ~(Xv | Ph);
    *MvOut = Ph & Xv;
    return hout;
}

static void cell_scores(Word P, Word M, int score, int out[WORD_SIZE]) /* edlib_ex.c:22-34 */
{
    Word mask = HIGH_BIT;
    for (int i = 0; i < WORD_SIZE - 1; ++i) {
        out[i] = score;
        if (P & mask) --score;
        if (M & mask) ++score;
        mask >>= 1;
    }
    out[WORD_SIZE - 1] = score;
}

static void ends_push(ora_aligner* a, int v)
{
    if (a->nends == a->mends) { a->mends = a->mends ? a->mends * 2 : 64; a->ends = xrealloc(a->ends, sizeof(int) * a->mends); }
    a->ends[a->nends++] = v;
}

/* edlib_ex.c:108-223, mode SHW only (the only mode Edlib_align uses) */
static int shw_distance(ora_aligner* a, int qn, const uint8_t* t, int tn, int k)
{
    a->nends = 0;
    const int nblk = nwords(qn), W = nblk * WORD_SIZE - qn;
    int fblk = 0, lblk = ORA_MIN(nwords(k + 1), nblk) - 1;
    Word *P = a->bP, *M = a->bM; int* S = a->bS;
    for (int b = 0; b <= lblk; ++b) { S[b] = (b + 1) * WORD_SIZE; P[b] = (Word)-1; M[b] = 0; }
    int best = -1;
    for (int c = 0; c < tn; ++c) {
        const Word* cpeq = a->peq + t[c] * ORA_MAXW;
        int hout = 1;
        for (int b = fblk; b <= lblk; ++b) { hout = advance_block(P[b], M[b], cpeq[b], hout, &P[b], &M[b]); S[b] += hout; }
        if (lblk < nblk - 1 && S[lblk] - hout <= k && ((cpeq[lblk + 1] & 1ULL) || hout < 0)) {
            ++lblk;
            P[lblk] = (Word)-1; M[lblk] = 0;
            int nh = advance_block(P[lblk], M[lblk], cpeq[lblk], hout, &P[lblk], &M[lblk]);
            S[lblk] = S[lblk - 1] - hout + WORD_SIZE + nh;
        } else {
            while (lblk >= fblk && S[lblk] >= k + WORD_SIZE) --lblk;
        }
        while (fblk <= lblk && S[fblk] >= k + WORD_SIZE) ++fblk;
        if (lblk < fblk) return best;
        if (lblk == nblk - 1) {
            int cs = S[lblk];
            if (cs <= k && (best == -1 || cs <= best)) {
                if (cs != best) { a->nends = 0; best = cs; k = best; }
                ends_push(a, c - W);
            }
        }
    }
    if (lblk == nblk - 1) {
        int sc[WORD_SIZE];
        cell_scores(P[lblk], M[lblk], S[lblk], sc);
        for (int i = 0; i < W; ++i) {
            int cs = sc[i + 1];
            if (cs <= k && (best == -1 || cs <= best)) {
                if (cs != best) { a->nends = 0; k = best = cs; }
                ends_push(a, tn - W + i);
            }
        }
    }
    return best;
}

/* edlib_ex.c:226-370 with traceback = TRUE, target_stop_position = -1 */
static int nw_distance(ora_aligner* a, int qn, const uint8_t* t, int tn, int k, int* end_position)
{
    *end_position = -1;
    if (k < abs(tn - qn)) return -1;
    k = ORA_MIN(k, ORA_MAX(qn, tn));
    const int nblk = nwords(qn), W = nblk * WORD_SIZE - qn;
    int fblk = 0, lblk;
    { int X = (k + qn - tn) / 2; int Y = ORA_MIN(k, X); lblk = ORA_MIN(nblk, nwords(Y + 1)) - 1; }
    Word *P = a->bP, *M = a->bM; int* S = a->bS;
    for (int b = 0; b <= lblk; ++b) { S[b] = (b + 1) * WORD_SIZE; P[b] = (Word)-1; M[b] = 0; }
    for (int c = 0; c < tn; ++c) {
        const Word* cpeq = a->peq + t[c] * ORA_MAXW;
        int hout = 1;
        for (int b = fblk; b <= lblk; ++b) { hout = advance_block(P[b], M[b], cpeq[b], hout, &P[b], &M[b]); S[b] += hout; }
        {
            int X1 = tn - c - 1, X2 = qn - ((1 + lblk) * WORD_SIZE - 1) - 1;
            int Z = ORA_MAX(X1, X2) + ((lblk == nblk - 1) ? W : 0) + S[lblk];
            k = ORA_MIN(k, Z);
        }
        if (lblk + 1 < nblk) {
            int r = (lblk + 1) * WORD_SIZE - 1 > k - S[lblk] + 2 * WORD_SIZE - 2 - tn + c + qn;
            if (!r) {
                ++lblk;
                P[lblk] = (Word)-1; M[lblk] = 0;
                int nh = advance_block(P[lblk], M[lblk], cpeq[lblk], hout, &P[lblk], &M[lblk]);
                S[lblk] = S[lblk - 1] - hout + WORD_SIZE + nh;
                hout = nh;
            }
        }
        while (lblk >= fblk && (S[lblk] >= k + WORD_SIZE ||
               ((lblk + 1) * WORD_SIZE - 1 > k - S[lblk] + 2 * WORD_SIZE - 2 - tn + c + qn + 1))) --lblk;
        while (fblk <= lblk && (S[fblk] >= k + WORD_SIZE ||
               ((fblk + 1) * WORD_SIZE - 1 < S[fblk] - k - tn + qn + c))) ++fblk;
        if (lblk < fblk) return -1;
        for (int b = fblk; b <= lblk; ++b) {
            a->Ps[(size_t)c * ORA_MAXW + b] = P[b]; a->Ms[(size_t)c * ORA_MAXW + b] = M[b]; a->Sc[(size_t)c * ORA_MAXW + b] = S[b];
        }
        a->first[c] = fblk; a->last[c] = lblk;
    }
    if (lblk == nblk - 1) {
        int sc[WORD_SIZE];
        cell_scores(P[lblk], M[lblk], S[lblk], sc);
        if (sc[W] <= k) { *end_position = tn - 1; return sc[W]; }
    }
    return -1;
}

enum { OP_MATCH = 0, OP_INS = 1, OP_DEL = 2, OP_MISMATCH = 3 }; /* edlib_ex.c:10-13 */

static void op_push(ora_aligner* a, unsigned char op)
{
    if (a->nops == a->mops) { a->mops = a->mops ? a->mops * 2 : 4096; a->ops = xrealloc(a->ops, a->mops); }
    a->ops[a->nops++] = op;
}

/* edlib_ex.c:383-621 obtainAlignmentTraceback; move priority up > left > diagonal */
static void traceback(ora_aligner* a, int qn, int tn, int bestScore)
{
    const int nblk = nwords(qn), W = nblk * WORD_SIZE - qn;
    a->nops = 0;
    int c = tn - 1, b = nblk - 1;
    int cur = bestScore, lS = -1, uS = -1, ulS = -1;
#define PS(c_, b_) a->Ps[(size_t)(c_) * ORA_MAXW + (b_)]
#define MS(c_, b_) a->Ms[(size_t)(c_) * ORA_MAXW + (b_)]
#define SC(c_, b_) a->Sc[(size_t)(c_) * ORA_MAXW + (b_)]
#define INBAND(c_, b_) ((b_) >= a->first[c_] && (b_) <= a->last[c_])
    Word curP = PS(c, b), curM = MS(c, b);
    int left = c > 0 && INBAND(c - 1, b);
    Word lP = 0, lM = 0;
    if (left) { lP = PS(c - 1, b); lM = MS(c - 1, b); }
    curP <<= W; curM <<= W;
    int pos = WORD_SIZE - W - 1;
    for (;;) {
        if (c == 0) { left = 1; lS = b * WORD_SIZE + pos + 1; ulS = lS - 1; }
        if (lS == -1 && left) {
            lS = SC(c - 1, b);
            for (int i = 0; i < WORD_SIZE - pos - 1; i++) {
                if (lP & HIGH_BIT) lS--;
                if (lM & HIGH_BIT) lS++;
                lP <<= 1; lM <<= 1;
            }
        }
        if (ulS == -1) {
            if (lS != -1) {
                ulS = lS;
                if (lP & HIGH_BIT) ulS--;
                if (lM & HIGH_BIT) ulS++;
            } else if (c > 0 && INBAND(c - 1, b - 1)) {
                ulS = SC(c - 1, b - 1);
            }
        }
        if (uS == -1) {
            uS = cur;
            if (curP & HIGH_BIT) uS--;
            if (curM & HIGH_BIT) uS++;
            curP <<= 1; curM <<= 1;
        }
        if (uS != -1 && uS + 1 == cur) {            /* up: consumes a query base */
            cur = uS; lS = ulS; uS = ulS = -1;
            if (pos == 0) {
                if (b == 0) {
                    op_push(a, OP_INS);
                    for (int i = 0; i < c + 1; ++i) op_push(a, OP_DEL);
                    break;
                } else {
                    pos = WORD_SIZE - 1; b--;
                    curP = PS(c, b); curM = MS(c, b);
                    if (c > 0 && INBAND(c - 1, b)) { left = 1; lP = PS(c - 1, b); lM = MS(c - 1, b); }
                    else left = 0;
                }
            } else { pos--; lP <<= 1; lM <<= 1; }
            op_push(a, OP_INS);
        } else if (lS != -1 && lS + 1 == cur) {      /* left: consumes a target base */
            cur = lS; uS = ulS; lS = ulS = -1;
            c--;
            if (c == -1) {
                op_push(a, OP_DEL);
                int numUp = b * WORD_SIZE + pos + 1;
                for (int i = 0; i < numUp; ++i) op_push(a, OP_INS);
                break;
            }
            curP = lP; curM = lM;
            if (c > 0 && INBAND(c - 1, b)) { left = 1; lP = PS(c - 1, b); lM = MS(c - 1, b); }
            else if (c == 0) { left = 1; lS = b * WORD_SIZE + pos + 1; ulS = lS - 1; }
            else left = 0;
            op_push(a, OP_DEL);
        } else if (ulS != -1) {                      /* diagonal */
            unsigned char mv = ulS == cur ? OP_MATCH : OP_MISMATCH;
            cur = ulS; uS = lS = ulS = -1;
            c--;
            if (c == -1) {
                op_push(a, mv);
                int numUp = b * WORD_SIZE + pos;
                for (int i = 0; i < numUp; ++i) op_push(a, OP_INS);
                break;
            }
            if (pos == 0) {
                if (b == 0) {
                    op_push(a, mv);
                    for (int i = 0; i < c + 1; ++i) op_push(a, OP_DEL);
                    break;
                }
                pos = WORD_SIZE - 1; b--;
                curP = PS(c, b); curM = MS(c, b);
            } else {
                pos--;
                curP = lP; curM = lM;
                curP <<= 1; curM <<= 1;
            }
            if (c > 0 && INBAND(c - 1, b)) { left = 1; lP = PS(c - 1, b); lM = MS(c - 1, b); }
            else if (c == 0) { left = 1; lS = b * WORD_SIZE + pos + 1; ulS = lS - 1; }
            else left = 0;
            op_push(a, mv);
        } else break;
    }
#undef PS
#undef MS
#undef SC
#undef INBAND
    for (size_t i = 0, j = a->nops; i + 1 < j; ++i) { --j; unsigned char x = a->ops[i]; a->ops[i] = a->ops[j]; a->ops[j] = x; }
}

/* edlib_ex.c:733-800 Edlib_align (+ :624-731 cigar -> gapped strings, folded: an 'M' run copies
 * both bases, 'I' = query base vs '-', 'D' = '-' vs target base) */
int ora_edlib_align(ora_aligner* a, const uint8_t* query, int qn, const uint8_t* target, int tn,
                    char* qaln, char* taln, int* qend, int* tend, int* edit_distance)
{
    qaln[0] = 0; taln[0] = 0; *qend = 0; *tend = 0;
    if (edit_distance) *edit_distance = -1;
    build_peq(query, qn, a->peq);
    int k = (int)(ORA_MIN(qn, tn) * a->error * 1.1);
    int d = shw_distance(a, qn, target, tn, k);
    if (d == -1) return 0;
    int endc = a->ends[0], endp;
    int d2 = nw_distance(a, qn, target, endc + 1, d, &endp);
    if (d2 != d || endp != endc) { fprintf(stderr, "oracle: NW/SHW disagree (%d/%d, %d/%d)\n", d2, d, endp, endc); abort(); }
    traceback(a, qn, endc + 1, d2);
    static const char dec[] = "ACGT-";
    int ai = 0, qi = 0, ti = 0;
    for (size_t i = 0; i < a->nops; ++i) {
        switch (a->ops[i]) {
        case OP_MATCH: case OP_MISMATCH: qaln[ai] = dec[query[qi++]]; taln[ai] = dec[target[ti++]]; break;
        case OP_INS: qaln[ai] = dec[query[qi++]]; taln[ai] = '-'; break;
        default: qaln[ai] = '-'; taln[ai] = dec[target[ti++]]; break;
        }
        ++ai;
    }
    qaln[ai] = 0; taln[ai] = 0;
    *qend = qi; *tend = ti;
    if (edit_distance) *edit_distance = d;
    return 1;
}

/* ===================================================================== onc_align */

static const int kMatCnt = 8; /* oc_aligner.c:9 */

/* oc_aligner.c:111-155 */
static int next_block(const uint8_t* query, int qidx, int qsize, const uint8_t* target, int tidx, int tsize,
                      int desired, int right, uint8_t* qfrag, int* qn, uint8_t* tfrag, int* tn)
{
    int last, qleft = qsize - qidx, tleft = tsize - tidx, qblk, tblk;
    if (qleft < desired + 100 || tleft < desired + 100) {
        qblk = (int)(tleft * 1.3); qblk = ORA_MIN(qblk, qleft);
        tblk = (int)(qleft * 1.3); tblk = ORA_MIN(tblk, tleft);
        last = 1;
    } else { qblk = desired; tblk = desired; last = 0; }
    if (right) {
        for (int i = 0; i < qblk; ++i) qfrag[i] = query[qidx + i];
        for (int i = 0; i < tblk; ++i) tfrag[i] = target[tidx + i];
    } else {
        for (int i = 0; i < qblk; ++i) qfrag[i] = query[-qidx - i];
        for (int i = 0; i < tblk; ++i) tfrag[i] = target[-tidx - i];
    }
    *qn = qblk; *tn = tblk;
    return last;
}

/* oc_aligner.c:157-286 */
static void oca_extend(ora_aligner* a, const uint8_t* query, int query_size, const uint8_t* target, int target_size,
                       int block_size, int right, ostr* qaln, ostr* taln, int tail_match_len)
{
    static const char dec[] = "ACGT-";
    ostr_clear(qaln); ostr_clear(taln);
    int qidx = 0, tidx = 0;
    for (;;) {
        int qfae, tfae, qn, tn;
        int last = next_block(query, qidx, query_size, target, tidx, target_size, block_size, right, a->qfrag, &qn, a->tfrag, &tn);
        if (qn == 0 || tn == 0) break;
        ora_edlib_align(a, a->qfrag, qn, a->tfrag, tn, a->qabuf, a->tabuf, &qfae, &tfae, NULL);
        int done = last;
        int acnt = 0, qcnt = 0, tcnt = 0;
        if (qn - qfae > 30 && tn - tfae > 30) done = 1;
        const int M = done ? tail_match_len : kMatCnt;
        int align_size = (int)strlen(a->qabuf);
        int k = align_size - 1, mm = 0;
        while (k >= 0) {
            char qc = a->qabuf[k], tc = a->tabuf[k];
            if (qc != '-') ++qcnt;
            if (tc != '-') ++tcnt;
            if (qc == tc) ++mm; else mm = 0;
            ++acnt;
            if (mm == M) break;
            --k;
        }
        if (mm != M || k < 1) {
            align_size = 0;
            for (int i = 0; i < qn && i < tn; ++i) {
                if (a->qfrag[i] != a->tfrag[i]) break;
                a->qabuf[align_size] = dec[a->qfrag[i]]; a->tabuf[align_size] = dec[a->tfrag[i]];
                ++align_size;
            }
            done = 1;
        } else {
            align_size -= acnt;
            qidx += qfae - qcnt; tidx += tfae - tcnt;
            if (done) align_size += M;
        }
        ostr_append(qaln, a->qabuf, (size_t)align_size);
        ostr_append(taln, a->tabuf, (size_t)align_size);
        if (done) break;
    }
}

/* oc_aligner.c:303-451 */
int ora_onc_align(ora_aligner* a, const uint8_t* query, int query_start, int query_size,
                  const uint8_t* target, int target_start, int target_size,
                  int block_size, int min_align_size, int tail_match_len, ora_align_result* res)
{
    ostr_clear(&a->qa); ostr_clear(&a->ta);
    int QS = query_start, TS = target_start;
    /* left extension: walks backwards from (QS-1, TS-1), strings are anchor-first */
    oca_extend(a, query + QS - 1, QS, target + TS - 1, TS, block_size, 0, &a->rq, &a->rt, tail_match_len);
    int rqcnt = 0, rtcnt = 0;
    {
        int qcnt = 0, tcnt = 0, acnt = 0, mm = 0;
        int rn = (int)a->rq.n;
        for (int i = 0; i < rn; ++i) {
            char qc = a->rq.s[i], tc = a->rt.s[i];
            if (qc != '-') ++qcnt;
            if (tc != '-') ++tcnt;
            if (qc == tc) ++mm; else mm = 0;
            ++acnt;
            if (mm == kMatCnt) break;
        }
        if (mm == kMatCnt) {
            QS -= qcnt; TS -= tcnt;
            for (int i = rn; i > acnt; --i) {
                char qc = a->rq.s[i - 1], tc = a->rt.s[i - 1];
                ostr_putc(&a->qa, qc); if (qc != '-') ++rqcnt;
                ostr_putc(&a->ta, tc); if (tc != '-') ++rtcnt;
            }
        }
    }
    oca_extend(a, query + QS, query_size - QS, target + TS, target_size - TS, block_size, 1, &a->fq, &a->ft, tail_match_len);
    int fqcnt = 0, ftcnt = 0;
    int fn = (int)a->fq.n;
    if (a->qa.n == 0) {
        int qcnt = 0, tcnt = 0, acnt = 0, mm = 0;
        for (int i = 0; i < fn; ++i) {
            char qc = a->fq.s[i], tc = a->ft.s[i];
            if (qc != '-') ++qcnt;
            if (tc != '-') ++tcnt;
            if (qc == tc) ++mm; else mm = 0;
            ++acnt;
            if (mm == kMatCnt) break;
        }
        if (mm == kMatCnt) {
            acnt -= kMatCnt; qcnt -= kMatCnt; tcnt -= kMatCnt;
            QS += qcnt; TS += tcnt;
            for (int i = acnt; i < fn; ++i) {
                char qc = a->fq.s[i], tc = a->ft.s[i];
                if (qc != '-') ++fqcnt;
                if (tc != '-') ++ftcnt;
                ostr_putc(&a->qa, qc); ostr_putc(&a->ta, tc);
            }
        }
    } else {
        for (int i = 0; i < fn; ++i) {
            char qc = a->fq.s[i], tc = a->ft.s[i];
            if (qc != '-') ++fqcnt;
            if (tc != '-') ++ftcnt;
            ostr_putc(&a->qa, qc); ostr_putc(&a->ta, tc);
        }
    }
    res->qoff = QS - rqcnt; res->qend = QS + fqcnt;
    res->toff = TS - rtcnt; res->tend = TS + ftcnt;
    int align_size = (int)a->ta.n;
    int nmat = 0;
    for (int i = 0; i < align_size; ++i) if (a->qa.s[i] == a->ta.s[i]) ++nmat;
    res->ident_perc = align_size ? 100.0 * nmat / align_size : 0.0; /* oc_aligner.c:288-301 */
    res->align_size = align_size;
    res->query_align = a->qa.s ? a->qa.s : "";
    res->target_align = a->ta.s ? a->ta.s : "";
    return align_size >= min_align_size;
}

/* ===================================================================== records + stage driver */

void ora_pack_candidate(const ora_candidate* c, uint32_t item[7]) /* gapped_candidate.c:13-30 */
{
    memset(item, 0, 28);
    if (c->sdir == 1) item[0] |= 1u << 31;
    if (c->qdir == 1) item[0] |= 1u << 30;
    if (c->qoff == c->qbeg) item[0] |= 1u << 29;
    item[0] |= (uint32_t)ORA_MIN(1000000, c->score);
    item[1] = (uint32_t)c->sid; item[2] = (uint32_t)c->sbeg; item[3] = (uint32_t)c->send;
    item[4] = (uint32_t)c->qid; item[5] = (uint32_t)c->qbeg; item[6] = (uint32_t)c->qend;
}

typedef struct { char* s; size_t n, m; } obuf;
static void obuf_put(obuf* o, const void* src, size_t len)
{
    if (o->n + len + 1 > o->m) { o->m = (o->n + len + 1) * 2; if (o->m < 4096) o->m = 4096; o->s = xrealloc(o->s, o->m); }
    memcpy(o->s + o->n, src, len); o->n += len;
}

typedef struct {
    const ora_options* opt;
    const ora_volume *reads, *ref;
    const ora_index* ix;
    int read_start_id, ref_start_id;
    int* next_chunk; pthread_mutex_t* lock;
    int nchunks; obuf* chunk_out;          /* [nchunks] */
    uint64_t n_records, aligned_q;
} worker_arg;

static const int kChunk = 500; /* pm_worker.c:13 / :354 */

/* pm_worker.c:29-83 extend_candidates (cans already sorted) */
static void extend_candidates(ora_candidate* cans, int ncan, ora_aligner* al, const uint8_t* fwd, const uint8_t* rev,
                              uint8_t** subject, size_t* msubject, const ora_volume* ref, int min_align,
                              ora_m4** m4s, size_t* nm4, size_t* mm4)
{
    *nm4 = 0;
    for (int i = 0; i < ncan; ++i) {
        ora_candidate* c = cans + i;
        int contained = 0;                                  /* map_aux.c:4-20 */
        for (size_t j = 0; j < *nm4; ++j) {
            ora_m4* m = *m4s + j;
            if (c->qdir == m->qdir && c->sid == m->sid &&
                (uint64_t)c->qoff >= m->qoff && (uint64_t)c->qoff <= m->qend &&
                (uint64_t)c->soff >= m->soff && (uint64_t)c->soff <= m->send) { contained = 1; break; }
        }
        if (contained) continue;
        const uint8_t* read = c->qdir == 0 ? fwd : rev;
        if ((size_t)c->ssize + 1 > *msubject) { *msubject = (size_t)c->ssize * 2 + 64; *subject = xrealloc(*subject, *msubject); }
        ora_volume_extract(ref, (uint64_t)c->sid, 0, *subject);
        ora_align_result r;
        if (ora_onc_align(al, read, (int)c->qoff, (int)c->qsize, *subject, (int)c->soff, (int)c->ssize,
                          512 /* kOcaBlockSize, edlib_ex_aux.h:23 */, min_align, 1 /* ONC_TAIL_MATCH_LEN_SHORT */, &r)) {
            ora_m4 m; memset(&m, 0, sizeof m);
            m.qid = c->qid; m.sid = c->sid; m.ident_perc = r.ident_perc; m.vscore = c->score; m.qdir = c->qdir;
            m.qoff = (uint64_t)r.qoff; m.qend = (uint64_t)r.qend; m.qext = (uint64_t)c->qoff; m.qsize = (uint64_t)c->qsize;
            m.sdir = 0; m.soff = (uint64_t)r.toff; m.send = (uint64_t)r.tend; m.sext = (uint64_t)c->soff; m.ssize = (uint64_t)c->ssize;
            if (m.qdir == 1) { uint64_t qo = m.qsize - m.qend, qe = m.qsize - m.qoff; m.qoff = qo; m.qend = qe; }
            if (*nm4 == *mm4) { *mm4 = *mm4 ? *mm4 * 2 : 64; *m4s = xrealloc(*m4s, *mm4 * sizeof(ora_m4)); }
            (*m4s)[(*nm4)++] = m;
        }
    }
}

static void* worker(void* arg_) /* pm_worker.c:85-204 */
{
    worker_arg* A = arg_;
    const ora_options* opt = A->opt;
    ora_wfd* w = ora_wfd_new(A->ref->nbases, opt->block_size, opt->kmer_size, opt->block_score_cutoff);
    ora_aligner* al = ora_aligner_new(opt->error);
    ora_can_vec cans = {0, 0, 0};
    uint8_t *fwd = NULL, *rev = NULL, *subject = NULL; size_t mread = 0, msubject = 0;
    ora_m4* m4s = NULL; size_t nm4 = 0, mm4 = 0;
    char line[1024];
    for (;;) {
        pthread_mutex_lock(A->lock);
        int ch = (*A->next_chunk)++;
        pthread_mutex_unlock(A->lock);
        if (ch >= A->nchunks) break;
        obuf* out = A->chunk_out + ch;
        int sid = ch * kChunk, eid = ORA_MIN((int)A->reads->nseq, sid + kChunk);
        for (int i = sid; i < eid; ++i) {
            size_t L = A->reads->size[i];
            if (L + 1 > mread) { mread = L * 2 + 64; fwd = xrealloc(fwd, mread); rev = xrealloc(rev, mread); }
            cans.n = 0;
            ora_volume_extract(A->reads, (uint64_t)i, 0, fwd);
            ora_find_candidates(fwd, (int)L, i, 0, A->read_start_id, A->ref_start_id, 1, A->ref, A->ix, opt, w, &cans);
            ora_volume_extract(A->reads, (uint64_t)i, 1, rev);
            ora_find_candidates(rev, (int)L, i, 1, A->read_start_id, A->ref_start_id, 1, A->ref, A->ix, opt, w, &cans);
            if (opt->job == 1) {
                qsort(cans.a, cans.n, sizeof(ora_candidate), cmp_can_pm);
                if (cans.n > (size_t)opt->num_candidates) cans.n = (size_t)opt->num_candidates;
                extend_candidates(cans.a, (int)cans.n, al, fwd, rev, &subject, &msubject, A->ref, opt->align_size_cutoff, &m4s, &nm4, &mm4);
                for (size_t k = 0; k < nm4; ++k) {
                    ora_m4* m = m4s + k;
                    A->aligned_q += m->qend - m->qoff;
                    int lq = m->qid, ls = m->sid;
                    m->qid += A->read_start_id; m->sid += A->ref_start_id;
                    if (opt->binary_output) { obuf_put(out, m, sizeof(ora_m4)); }
                    else {
                        int n;
                        if (opt->use_hdr_as_id)     /* m4_record.h:99-124 */
                            n = snprintf(line, sizeof line, "%s\t%s\t%.2f\t%d\t%d\t%lu\t%lu\t%lu\t%d\t%lu\t%lu\t%lu\n",
                                         A->reads->hdr + A->reads->hdr_offset[lq], A->ref->hdr + A->ref->hdr_offset[ls],
                                         m->ident_perc, m->vscore, m->qdir, m->qoff, m->qend, m->qsize, m->sdir, m->soff, m->send, m->ssize);
                        else                        /* m4_record.h:72-97 */
                            n = snprintf(line, sizeof line, "%d\t%d\t%.2f\t%d\t%d\t%lu\t%lu\t%lu\t%d\t%lu\t%lu\t%lu\n",
                                         m->qid, m->sid, m->ident_perc, m->vscore, m->qdir, m->qoff, m->qend, m->qsize, m->sdir, m->soff, m->send, m->ssize);
                        obuf_put(out, line, (size_t)n);
                    }
                    ++A->n_records;
                }
            } else {
                for (size_t k = 0; k < cans.n; ++k) { cans.a[k].qid += A->read_start_id; cans.a[k].sid += A->ref_start_id; }
                if (cans.n > (size_t)opt->num_candidates) {
                    qsort(cans.a, cans.n, sizeof(ora_candidate), cmp_can_pm);
                    cans.n = (size_t)opt->num_candidates;
                }
                for (size_t k = 0; k < cans.n; ++k) {
                    ora_candidate* c = cans.a + k;
                    if (opt->binary_output) { uint32_t item[7]; ora_pack_candidate(c, item); obuf_put(out, item, 28); }
                    else {                           /* gapped_candidate.h:26-42 */
                        int n = snprintf(line, sizeof line, "%d\t%d\t%d\t%d\t%lu\t%lu\t%lu\t%lu\t%d\t%lu\t%lu\t%lu\t%lu\n",
                                         c->qid, c->sid, c->score, c->qdir, (unsigned long)c->qbeg, (unsigned long)c->qend,
                                         (unsigned long)c->qoff, (unsigned long)c->qsize, c->sdir, (unsigned long)c->sbeg,
                                         (unsigned long)c->send, (unsigned long)c->soff, (unsigned long)c->ssize);
                        obuf_put(out, line, (size_t)n);
                    }
                    ++A->n_records;
                }
            }
        }
    }
    free(fwd); free(rev); free(subject); free(m4s); free(cans.a);
    ora_wfd_free(w); ora_aligner_free(al);
    return NULL;
}

int ora_pm_main(const ora_options* opt, int vid, const char* wrk_dir, const char* output, ora_stats* stats)
{   /* pm_worker.c:338-400.  Unlike the reference, records are written in read order. */
    ora_stats st; memset(&st, 0, sizeof st);
    ora_volumes_info vi;
    if (ora_volumes_info_load(wrk_dir, &vi)) { fprintf(stderr, "oracle: cannot load volume info from %s\n", wrk_dir); return 1; }
    if (vid < 0 || vid >= vi.num_volumes) { fprintf(stderr, "oracle: bad volume id %d\n", vid); return 1; }
    ora_volume ref;
    if (ora_volume_load(vi.names[vid], &ref)) { fprintf(stderr, "oracle: cannot load %s\n", vi.names[vid]); return 1; }
    double t0 = now_sec();
    ora_index* ix = ora_index_build(&ref, opt->kmer_size, opt->kmer_cnt_cutoff);
    st.t_index = now_sec() - t0;
    FILE* out = fopen(output, "w");
    if (!out) { fprintf(stderr, "oracle: cannot open %s\n", output); return 1; }
    int nt = opt->num_threads > 0 ? opt->num_threads : 1;
    for (int v = vid; v < vi.num_volumes; ++v) {
        ora_volume reads_own; const ora_volume* reads = &ref;
        if (v != vid) { if (ora_volume_load(vi.names[v], &reads_own)) return 1; reads = &reads_own; }
        double t1 = now_sec();
        int nchunks = (int)((reads->nseq + kChunk - 1) / kChunk), next = 0;
        obuf* cout = xcalloc((size_t)nchunks, sizeof(obuf));
        pthread_mutex_t lock; pthread_mutex_init(&lock, NULL);
        worker_arg* args = xcalloc((size_t)nt, sizeof(worker_arg));
        pthread_t* th = xcalloc((size_t)nt, sizeof(pthread_t));
        for (int t = 0; t < nt; ++t) {
            worker_arg a = { opt, reads, &ref, ix, vi.read_start_id[v], vi.read_start_id[vid], &next, &lock, nchunks, cout, 0, 0 };
            args[t] = a;
            pthread_create(th + t, NULL, worker, args + t);
        }
        for (int t = 0; t < nt; ++t) { pthread_join(th[t], NULL); st.n_records += args[t].n_records; st.aligned_qbases += args[t].aligned_q; }
        st.t_map += now_sec() - t1;
        for (int c = 0; c < nchunks; ++c) { if (cout[c].n) fwrite(cout[c].s, 1, cout[c].n, out); free(cout[c].s); }
        free(cout); free(args); free(th);
        if (v != vid) ora_volume_free(&reads_own);
    }
    fclose(out);
    ora_index_free(ix); ora_volume_free(&ref); ora_volumes_info_free(&vi);
    if (stats) *stats = st;
    return 0;
}
