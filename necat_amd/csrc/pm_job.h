// pm_job.h - one oc2pmov job (pm_worker.c:338 pm_main): reference volume `vid` against the volumes vid .. V-1, records to `output`.
// Shared by oc2pmov (one job per process, as necat.pl launches it) and oc2pm (one resident process per GPU runs the jobs of all
// its volumes on one context: no HIP start-up, no pool allocation per volume).  While volume i is on the GPU a host thread
// already reads volume i + 1 from disk.
#pragma once
#include <future>
#include <memory>
#include "host_io.h"
#include "host_fmt.h"

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

// `preloaded` (optional): the reference volume, already being read (oc2pmov reads it while the HIP runtime starts)
inline int pm_run_volume(necat_ctx* ctx, const VolumesInfo& vi, int vid, const necat_map_options& opt, const char* output, const char* tag, const PmTrace& tr,
                         std::future<std::unique_ptr<PmLoaded>>* preloaded = nullptr)
{
    auto fail = [&](const char* what, const char* detail) { fprintf(stderr, "[%s] ERROR: %s: %s\n", tag, what, detail); return 1; };
    std::string err;
    int rc;
    std::unique_ptr<PmLoaded> ref_l = preloaded ? preloaded->get() : pm_load_async(vi, vid).get();
    if (!ref_l->ok) return fail("volume", ref_l->err.c_str());
    HostVolume& href = ref_l->v;
    tr.stage("volume read");
    necat_volume* ref = nullptr;
    if ((rc = necat_volume_upload(ctx, href.pac.data(), href.nbases, href.offset.data(), href.size.data(), href.offset.size(), &ref)))
        return fail("necat_volume_upload", necat_last_error(ctx));
    tr.stage("volume uploaded");
    log_line("", "build_lookup_table");
    double t0 = now_sec();
    necat_index* ix = nullptr;
    if ((rc = necat_index_build(ctx, ref, opt.kmer_size, opt.kmer_cnt_cutoff, &ix))) return fail("necat_index_build", necat_last_error(ctx));
    log_line("[%s] INFO: '%s' takes %.2lf secs.\n", "build_lookup_table", now_sec() - t0);
    tr.stage("index built");

    // write to a temporary name first: a failed run never leaves a complete-looking pm_result_i
    const std::string tmp_out = std::string(output) + ".part";
    FILE* out = fopen(tmp_out.c_str(), "w");
    if (!out) return fail("output", "cannot open for writing");
    const int ref_start = vi.read_start_id[vid];
    const int pcan_batch = (opt.job == 0 && getenv("NECAT_PM_PARTITIONS")) ? atoi(getenv("NECAT_PM_PARTITIONS")) : 0;
    if (pcan_batch > 0)      // this job's partition files are appended to: start from nothing
        for (int p = 0; p < (vi.num_reads + pcan_batch - 1) / pcan_batch; ++p) remove((std::string(output) + ".p" + std::to_string(p)).c_str());
    // the next volume is read from disk while this one is mapped
    typedef PmLoaded Loaded;
    auto load_async = [&](int i) { return pm_load_async(vi, i); };
    std::future<std::unique_ptr<Loaded>> next;
    if (vid + 1 < vi.num_volumes) next = load_async(vid + 1);
    int status = 0;
    for (int i = vid; i < vi.num_volumes && !status; ++i) {     // pm_worker.c:372-390
        char job[256];
        snprintf(job, sizeof job, "pairwise mapping v%d vs v%d", i, vid);
        log_line("", job);
        t0 = now_sec();
        std::unique_ptr<Loaded> own;
        const HostVolume* hreads = &href;
        necat_volume* reads = ref;
        if (i != vid) {
            own = next.get();
            if (i + 1 < vi.num_volumes) next = load_async(i + 1);
            if (!own->ok) { status = fail("volume", own->err.c_str()); break; }
            hreads = &own->v;
            if ((rc = necat_volume_upload(ctx, own->v.pac.data(), own->v.nbases, own->v.offset.data(), own->v.size.data(), own->v.offset.size(), &reads)))
                { status = fail("necat_volume_upload", necat_last_error(ctx)); break; }
        }
        const int read_start = vi.read_start_id[i];
        necat_candidate* cands = nullptr; uint64_t ncand = 0;
        bool wok = true;
        if (opt.job == 1) {
            // pm_search_one_volume with -j 1: seeding + extension, the candidates stay on the device
            necat_m4* m4 = nullptr; uint64_t nm4 = 0;
            if ((rc = necat_map_pair(ctx, ix, ref, reads, read_start, ref_start, 1, &opt, 1 /* ONC_TAIL_MATCH_LEN_SHORT */, &m4, &nm4, &ncand)))
                status = fail("necat_map_pair", necat_last_error(ctx));
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
                    wok = write_records(out, nm4, max_len, opt.num_threads, [&](char* p, uint64_t k) {
                        const necat_m4& m = m4[k];
                        return hdr ? put_m4(p, m, hreads->name((uint64_t)(m.qid - read_start)), href.name((uint64_t)(m.sid - ref_start))) : put_m4(p, m, nullptr, nullptr);
                    });
                }
                necat_free(m4);
            }
        } else {
            if ((rc = necat_find_candidates(ctx, ix, ref, reads, read_start, ref_start, 1, &opt, &cands, &ncand)))
                status = fail("necat_find_candidates", necat_last_error(ctx));
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
                    for (uint64_t k = 0; k < ncand; ++k) pack_candidate(&cands[k], items.data() + 7 * k);
                    wok = ncand == 0 || fwrite(items.data(), 28, ncand, out) == ncand;
                } else wok = write_records(out, ncand, 13 * 24, opt.num_threads, [&](char* p, uint64_t k) { return put_candidate(p, cands[k]); });     // DUMP_GAPPED_CANDIDATE (gapped_candidate.h:26-42)
            }
        }
        necat_free(cands);
        if (reads != ref) necat_volume_free(ctx, reads);
        if (!status && !wok) status = fail("output", "write failed");
        if (!status) { tr.stage("records written", i, vid); log_line("[%s] INFO: '%s' takes %.2lf secs.\n", job, now_sec() - t0); }
    }
    if (next.valid()) next.wait();
    if (fclose(out) != 0 && !status) status = fail("output", "write failed");
    if (!status && rename(tmp_out.c_str(), output) != 0) status = fail("output", "rename failed");
    if (status) remove(tmp_out.c_str());
    necat_index_free(ctx, ix);
    necat_volume_free(ctx, ref);
    tr.stage("job done");
    return status;
}

}  // namespace necat_host
