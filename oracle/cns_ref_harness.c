/* cns_ref_harness.c - TEST INFRASTRUCTURE ONLY (never part of the product).
 *
 * Runs the REFERENCE's own consensus driver (consensus/consensus_one_partition.c:128, one thread) over the
 * candidate partitions of a work directory and logs what its extension loop decides: every add_one_align
 * call (tasc/cbcns.c:47 - one accepted overlap with its gapped strings and weight) and, per template, the
 * numbers consensus_one_read leaves in CnsSeq (ident_cutoff, num_can, num_ovlps) and cov_ranges when it
 * hands over to consensus_broken / consensus_unbroken (consensus_one_read.c:376-392).
 *
 * The reference objects are linked untouched (oracle/_ref/libnecat_cns_ref.so, built by oracle/Makefile from
 * the sources under /root/reference); this file only INTERPOSES the three entry points of tasc/cbcns.c that
 * the loop calls: the definitions below win at dynamic-link time and forward to the real ones (RTLD_NEXT).
 *
 *   cns_ref_harness [oc2cns options] wrk_dir candidates_prefix log_out [full]
 *
 * Log lines (tab-separated):
 *   A  toff tend weight aln_size fnv(qaln) fnv(taln) [qaln taln]      (strings only with "full")
 *   T  template_id template_size ident_cutoff num_can num_ovlps n_ranges [first second]...
 * A lines precede the T line of their template.
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common/makedb_aux.h"
#include "consensus/cns_options.h"
#include "consensus/consensus_one_partition.h"
#include "partition_candidates/pcan_aux.h"
#include "tasc/cbcns.h"

static FILE* g_log = NULL;
static int g_full = 0;

static unsigned long long fnv(const char* s, size_t n)
{
    unsigned long long h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)s[i]; h *= 1099511628211ULL; }
    return h;
}

void add_one_align(CbCnsData* cns_data, const char* qaln, const char* taln, const size_t aln_size,
                   kstring_t* target, int toff, int tend, double weight)
{
    static void (*real)(CbCnsData*, const char*, const char*, const size_t, kstring_t*, int, int, double) = NULL;
    if (!real) real = dlsym(RTLD_NEXT, "add_one_align");
    fprintf(g_log, "A\t%d\t%d\t%.17g\t%zu\t%016llx\t%016llx", toff, tend, weight, aln_size, fnv(qaln, aln_size), fnv(taln, aln_size));
    if (g_full) fprintf(g_log, "\t%.*s\t%.*s", (int)aln_size, qaln, (int)aln_size, taln);
    fputc('\n', g_log);
    real(cns_data, qaln, taln, aln_size, target, toff, tend, weight);
}

static void log_template(int template_id, int template_size, const CnsSeq* cs, const vec_intpair* ranges)
{
    const size_t nr = ranges ? kv_size(*ranges) : 0;
    fprintf(g_log, "T\t%d\t%d\t%.17g\t%d\t%d\t%zu", template_id, template_size, cs->ident_cutoff, cs->num_can, cs->num_ovlps, nr);
    for (size_t i = 0; i < nr; ++i) fprintf(g_log, "\t%d\t%d", kv_A(*ranges, i).first, kv_A(*ranges, i).second);
    fputc('\n', g_log);
}

void consensus_broken(CbCnsData* cns_data, const int min_cov, const int min_size, const int template_id,
                      const int template_size, vec_intpair* cns_intvs, CnsSeq* cns_seq, RecordWriter* out,
                      const int check_chimeric_read, vec_intpair* cov_ranges)
{
    static void (*real)(CbCnsData*, const int, const int, const int, const int, vec_intpair*, CnsSeq*, RecordWriter*,
                        const int, vec_intpair*) = NULL;
    if (!real) real = dlsym(RTLD_NEXT, "consensus_broken");
    log_template(template_id, template_size, cns_seq, cov_ranges);
    real(cns_data, min_cov, min_size, template_id, template_size, cns_intvs, cns_seq, out, check_chimeric_read, cov_ranges);
}

int main(int argc, char* argv[])
{
    if (argc >= 2 && strcmp(argv[argc - 1], "full") == 0) { g_full = 1; --argc; }
    if (argc < 4) { fprintf(stderr, "usage: %s [oc2cns options] wrk_dir candidates_prefix log_out [full]\n", argv[0]); return 1; }
    CnsOptions options;
    if (parse_CnsOptions(argc - 3, argv, &options) != ARG_PARSE_SUCCESS) { fprintf(stderr, "bad options\n"); return 1; }
    options.num_threads = 1;            /* templates in partition order */
    if (options.full_consensus) { fprintf(stderr, "full consensus (-f 1) is not logged by this harness\n"); return 1; }
    const char* wrk_dir = argv[argc - 3];
    const char* can_path = argv[argc - 2];
    g_log = fopen(argv[argc - 1], "w");
    if (!g_log) { perror("log"); return 1; }
    PackedDB* reads = options.small_memory ? NULL : merge_volumes(wrk_dir);
    const int np = load_num_partitions(can_path);
    FILE* cns_out = fopen("/dev/null", "w");
    FILE* raw_out = fopen("/dev/null", "w");
    for (int i = 0; i < np; ++i) consensus_one_partition(wrk_dir, reads, can_path, &options, cns_out, raw_out, i);
    fclose(cns_out); fclose(raw_out); fclose(g_log);
    if (reads) free_PackedDB(reads);
    return 0;
}
