// pm_job.h - one oc2pmov job (pm_worker.c:338 pm_main): reference volume `vid` against the volumes vid .. V-1, records to `output`.
// Shared by oc2pmov (one job per process, as necat.pl launches it) and oc2pm (one resident process per GPU runs the jobs of all
// its volumes on one context: no HIP start-up, no pool allocation per volume).  While volume i is on the GPU a host thread
// already reads volume i + 1 from disk.
// A job can also be a SHARE of that job: a list of units (query volume, slot range) from the pair scheduler (pair_sched.h) - what
// oc2pm's workers run when several GPUs split the volume pairs of a project; the shares of all workers concatenate to the job's file.
#pragma once
#include <future>
#include <memory>
#include "host_io.h"
#include "host_fmt.h"
#include "pair_sched.h"

namespace necat_host {

struct PmTrace {       // NECAT_CLI_TRACE=1: wall clock of the stages on stderr
    double t0 = now_sec();
    bool on = getenv("NECAT_CLI_TRACE") && atoi(getenv("NECAT_CLI_TRACE"));
    void stage(const char* what, int a = -1, int b = -1) const
    {
        if (!on) return;
        if (a >= 0) fprintf(stderr, "[pm] %8.1f ms  %s (v%d vs v%d)\n", (now_sec() - t0) * 1e3, what, a, b);
        else fprintf(stderr, "[pm] %8.1f ms  %s\n", (now_sec() - t0) * 1e3, what);
    }
};

// returns 0, or 1 after printing "[tag] ERROR: ..." (every fatal error of the reference is OC_ERROR -> exit 1)
struct PmLoaded { HostVolume v; bool ok = false; std::string err; };
inline std::future<std::unique_ptr<PmLoaded>> pm_load_async(const VolumesInfo& vi, int i)
{
    return std::async(std::launch::async, [&vi, i]() { auto l = std::make_unique<PmLoaded>(); l->ok = load_volume(vi.names[i].c_str(), &l->v, &l->err); return l; });
}

// one unit of a job: query volume `query_vol`, of which the chunks c (pair_chunk_reads reads each) with slot_lo <= c % slots < slot_hi
struct PmUnit { int query_vol, slot_lo, slot_hi; };
constexpr int kPmSlots = 64;

// `preloaded` (optional): the reference volume, already being read (oc2pmov reads it while the HIP runtime starts)
// `units` (optional): this process's share of the job, query volumes ascending; nullptr = the whole job (every query volume vid .. V-1)
inline int pm_run_volume(necat_ctx* ctx, const VolumesInfo& vi, int vid, const necat_map_options& opt, const char* output, const char* tag, const PmTrace& tr,
                         std::future<std::unique_ptr<PmLoaded>>* preloaded = nullptr, const std::vector<PmUnit>* units = nullptr)
{
    auto fail = [&](const char* what, const char* detail) { fprintf(stderr, "[%s] ERROR: %s: %s\n", tag, what, detail); return 1; };
    std::vector<PmUnit> whole;
    if (!units) { for (int i = vid; i < vi.num_volumes; ++i) whole.push_back(PmUnit{i, 0, kPmSlots}); units = &whole; }
    // everything the job holds, released on EVERY way out (a resident oc2pm worker outlives a failed job)
    struct Held {
        necat_ctx* ctx; necat_volume* ref = nullptr; necat_index* ix = nullptr; necat_volume* reads = nullptr; FILE* out = nullptr;
        std::future<std::unique_ptr<PmLoaded>> next; std::string tmp_out;
        ~Held()
        {
            if (next.valid()) next.wait();
            if (out) { fclose(out); remove(tmp_out.c_str()); }
            if (reads && reads != ref) necat_volume_free(ctx, reads);
            if (ix) necat_index_free(ctx, ix);
            if (ref) necat_volume_free(ctx, ref);
        }
    } H{ctx};
    int rc;
    std::unique_ptr<PmLoaded> ref_l = preloaded ? preloaded->get() : pm_load_async(vi, vid).get();
    if (!ref_l->ok) return fail("volume", ref_l->err.c_str());
    HostVolume& href = ref_l->v;
    tr.stage("volume read");
    if ((rc = necat_volume_upload(ctx, href.pac.data(), href.nbases, href.offset.data(), href.size.data(), href.offset.size(), &H.ref)))
        return fail("necat_volume_upload", necat_last_error(ctx));
    necat_volume* const ref = H.ref;
    tr.stage("volume uploaded");
    log_line("", "build_lookup_table");
    double t0 = now_sec();
    if ((rc = necat_index_build(ctx, ref, opt.kmer_size, opt.kmer_cnt_cutoff, &H.ix))) return fail("necat_index_build", necat_last_error(ctx));
    necat_index* const ix = H.ix;
    log_line("[%s] INFO: '%s' takes %.2lf secs.\n", "build_lookup_table", now_sec() - t0);
    tr.stage("index built");

    // write to a temporary name first: a failed run never leaves a complete-looking pm_result_i
    H.tmp_out = std::string(output) + ".part";
    H.out = fopen(H.tmp_out.c_str(), "w");
    if (!H.out) return fail("output", "cannot open for writing");
    FILE* const out = H.out;
    const int ref_start = vi.read_start_id[vid];
    const int pcan_batch = (opt.job == 0 && getenv("NECAT_PM_PARTITIONS")) ? atoi(getenv("NECAT_PM_PARTITIONS")) : 0;
    if (pcan_batch > 0)      // this job's partition files are appended to: start from nothing
        for (int p = 0; p < (vi.num_reads + pcan_batch - 1) / pcan_batch; ++p) remove((std::string(output) + ".p" + std::to_string(p)).c_str());
    // the next query volume is read from disk while this one is mapped
    typedef PmLoaded Loaded;
    auto next_other = [&](size_t k) -> int {      // the first unit after unit k whose query volume is neither unit k's nor the reference
        for (size_t j = k + 1; j < units->size(); ++j) if ((*units)[j].query_vol != (*units)[k].query_vol && (*units)[j].query_vol != vid) return (*units)[j].query_vol;
        return -1;
    };
    int next_vol = -1, cur_vol = -1;
    auto prefetch = [&](int v) { if (v >= 0) { H.next = pm_load_async(vi, v); next_vol = v; } else next_vol = -1; };
    if (!units->empty()) {
        const int first = (*units)[0].query_vol != vid ? (*units)[0].query_vol : next_other(0);
        prefetch(first);
    }
    std::unique_ptr<Loaded> own;
    int status = 0;
    for (size_t k = 0; k < units->size() && !status; ++k) {     // pm_worker.c:372-390
        const PmUnit& u = (*units)[k];
        const int i = u.query_vol;
        if (i < vid || i >= vi.num_volumes || u.slot_lo < 0 || u.slot_hi > kPmSlots || u.slot_lo >= u.slot_hi) { status = fail("unit", "outside the job"); break; }
        const bool whole_pair = u.slot_lo == 0 && u.slot_hi == kPmSlots;
        const int chunk_reads = pair_chunk_reads((uint64_t)vi.read_count[i], kPmSlots);
        char job[256];
        if (whole_pair) snprintf(job, sizeof job, "pairwise mapping v%d vs v%d", i, vid);
        else snprintf(job, sizeof job, "pairwise mapping v%d vs v%d (query chunks %d..%d of %d)", i, vid, u.slot_lo, u.slot_hi - 1, kPmSlots);
        log_line("", job);
        t0 = now_sec();
        const HostVolume* hreads = &href;
        if (i == vid) {
            if (H.reads && H.reads != ref) necat_volume_free(ctx, H.reads);
            H.reads = ref; cur_vol = vid;
        } else if (i != cur_vol) {
            if (H.reads && H.reads != ref) { necat_volume_free(ctx, H.reads); H.reads = nullptr; }
            if (next_vol != i) { if (H.next.valid()) H.next.wait(); prefetch(i); }
            own = H.next.get();
            prefetch(next_other(k));
            if (!own->ok) { status = fail("volume", own->err.c_str()); break; }
            if ((rc = necat_volume_upload(ctx, own->v.pac.data(), own->v.nbases, own->v.offset.data(), own->v.size.data(), own->v.offset.size(), &H.reads)))
                { H.reads = nullptr; status = fail("necat_volume_upload", necat_last_error(ctx)); break; }
            cur_vol = i;
        }
        if (i != vid) hreads = &own->v;
        necat_volume* const reads = H.reads;
        const int read_start = vi.read_start_id[i];
        necat_candidate* cands = nullptr; uint64_t ncand = 0;
        bool wok = true;
        if (opt.job == 1) {
            // pm_search_one_volume with -j 1: seeding + extension, the candidates stay on the device
            necat_m4* m4 = nullptr; uint64_t nm4 = 0;
            rc = whole_pair ? necat_map_pair(ctx, ix, ref, reads, read_start, ref_start, 1, &opt, 1 /* ONC_TAIL_MATCH_LEN_SHORT */, &m4, &nm4, &ncand)
                            : necat_map_pair_part(ctx, ix, ref, reads, read_start, ref_start, 1, &opt, 1, chunk_reads, u.slot_lo, u.slot_hi, kPmSlots, &m4, &nm4, &ncand);
            if (rc) status = fail("necat_map_pair", necat_last_error(ctx));
            else {
                tr.stage("mapped", i, vid);
                if (opt.binary_output) wok = nm4 == 0 || fwrite(m4, sizeof(necat_m4), nm4, out) == nm4;
                else {
                    const bool hdr = opt.use_hdr_as_id != 0;      // DUMP_ASM_M4_HDR_ID (m4_record.h:99-124) / DUMP_ASM_M4 (:72-97)
                    size_t max_len = 12 * 24;
                    if (hdr) {
                        size_t lq = 0, ls = 0;
                        for (uint64_t r = 0; r < hreads->offset.size(); ++r) lq = std::max(lq, strlen(hreads->name(r)));
                        for (uint64_t r = 0; r < href.offset.size(); ++r) ls = std::max(ls, strlen(href.name(r)));
                        max_len += lq + ls;
                    }
                    wok = write_records(out, nm4, max_len, opt.num_threads, [&](char* p, uint64_t k2) {
                        const necat_m4& m = m4[k2];
                        return hdr ? put_m4(p, m, hreads->name((uint64_t)(m.qid - read_start)), href.name((uint64_t)(m.sid - ref_start))) : put_m4(p, m, nullptr, nullptr);
                    });
                }
                necat_free(m4);
            }
        } else {
            rc = whole_pair ? necat_find_candidates(ctx, ix, ref, reads, read_start, ref_start, 1, &opt, &cands, &ncand)
                            : necat_find_candidates_part(ctx, ix, ref, reads, read_start, ref_start, 1, &opt, chunk_reads, u.slot_lo, u.slot_hi, kPmSlots, &cands, &ncand);
            if (rc) status = fail("necat_find_candidates", necat_last_error(ctx));
            else {
                tr.stage("candidates found", i, vid);
                // NECAT_PM_PARTITIONS=<batch size>: the consensus stage's partitions (oc2pcan's candidates.p<i>) straight from the
                // candidates of this job, partitioned on the device (necat_pcan_partition) - the pipeline can then skip oc2pcan's
                // write + read of the whole candidates file
                if (pcan_batch > 0) {
                    uint32_t* recs = nullptr; uint64_t* poff = nullptr; int np = 0;
                    if ((rc = necat_pcan_partition(ctx, cands, ncand, pcan_batch, vi.num_reads, &recs, &poff, &np)))
                        status = fail("necat_pcan_partition", necat_last_error(ctx));
                    else {
                        for (int p = 0; p < np && wok; ++p) {
                            if (poff[p + 1] == poff[p]) continue;
                            FILE* pf = fopen((std::string(output) + ".p" + std::to_string(p)).c_str(), "ab");
                            wok = pf && fwrite(recs + 7 * poff[p], 28, poff[p + 1] - poff[p], pf) == poff[p + 1] - poff[p];
                            if (pf && fclose(pf) != 0) wok = false;
                        }
                        tr.stage("partitions written", i, vid);
                    }
                    necat_free(recs); necat_free(poff);
                }
                if (opt.binary_output) {
                    std::vector<uint32_t> items((size_t)ncand * 7);
                    for (uint64_t k2 = 0; k2 < ncand; ++k2) pack_candidate(&cands[k2], items.data() + 7 * k2);
                    wok = ncand == 0 || fwrite(items.data(), 28, ncand, out) == ncand;
                } else wok = write_records(out, ncand, 13 * 24, opt.num_threads, [&](char* p, uint64_t k2) { return put_candidate(p, cands[k2]); });     // DUMP_GAPPED_CANDIDATE (gapped_candidate.h:26-42)
            }
        }
        necat_free(cands);
        if (!status && !wok) status = fail("output", "write failed");
        if (!status) { tr.stage("records written", i, vid); log_line("[%s] INFO: '%s' takes %.2lf secs.\n", job, now_sec() - t0); }
    }
    if (H.next.valid()) H.next.wait();
    FILE* f = H.out; H.out = nullptr;                 // closed here: the guard only cleans up after a failure
    if (fclose(f) != 0 && !status) status = fail("output", "write failed");
    if (!status && rename(H.tmp_out.c_str(), output) != 0) status = fail("output", "rename failed");
    if (status) remove(H.tmp_out.c_str());
    tr.stage("job done");
    return status;
}

}  // namespace necat_host
