// pm_job.h - one oc2pmov job (pm_worker.c:338 pm_main): reference volume `vid` against the volumes vid .. V-1, records to `output`.
// Shared by oc2pmov (one job per process, as necat.pl launches it) and oc2pm (one resident process per GPU runs the jobs of all
// its volumes on one context: no HIP start-up, no pool allocation per volume).  While volume i is on the GPU a host thread
// already reads volume i + 1 from disk.
// A job can also be a SHARE of that job: a list of units (query volume, slot range) from the pair scheduler (pair_sched.h) - what
// oc2pm's workers run when several GPUs split the volume pairs of a project; the shares of all workers concatenate to the job's file.
#pragma once
#include <condition_variable>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
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

// Pair lanes (round 6): the contexts beside the job's own on which a job's units run side by side (pm_run_volume).  One unit's extension ends in ~ 15 rounds that are
// one block's dependent chain each on a mostly idle chip, and its seeding is HBM- / latency-bound while the DP kernels are issue-bound - two or three units in
// flight fill each other's gaps.  Warm, pairs in flight against one at a time (tools/r06/run9.sh, run23.sh, run24.sh): E. coli-size pairs 38.1 -> 32.0 -> 30.1 -> 29.8 ms
// per pair at 1 / 2 / 3 / 4; 0.6 Gbp SENSITIVE pairs 267 -> 257 -> 246 ms at 1 / 2 / 3; a 2 Gbp volume against itself 231 -> 202 ms (-j 1), 125 -> 115 ms (-j 0) at 1 / 2.
// What a lane costs is its context's arenas (tens of GB at 2 Gbp, mapped on first use): a 5.6 Gbp project of SIX pairs through one worker took 2.51 - 2.62 s at one lane,
// 2.71 - 2.82 at two, 2.84 - 2.94 at three (tools/r06/run11.sh) - the second context's first touches cost more than six pairs gain.
// NECAT_PAIR_LANES = 1 .. 8 fixes the number; unset = 2 lanes for a job whose reference volume is below kPmLaneBases bases (cheap arenas) or that has at least
// kPmLaneUnits units (enough pairs to pay for them: volume v of a 45-volume project has 45 - v), 1 otherwise.
// The contexts are made by their lane threads on first use (beside lane 0's first unit) and kept for the owner's next job.
constexpr uint64_t kPmLaneBases = 400000000ull;
constexpr size_t kPmLaneUnits = 8;
struct PmLanes {
    int device, lanes;
    bool fixed;
    std::vector<necat_ctx*> extra;
    explicit PmLanes(int dev) : device(dev)
    {
        const char* e = getenv("NECAT_PAIR_LANES");
        fixed = e && *e;
        lanes = fixed ? atoi(e) : 2;
        if (lanes < 1) lanes = 1;
        if (lanes > 8) lanes = 8;
        extra.assign((size_t)lanes - 1, nullptr);
    }
    int for_job(uint64_t ref_bases, size_t units) const { return fixed ? lanes : ((ref_bases < kPmLaneBases || units >= kPmLaneUnits) ? lanes : 1); }
    void close() { for (necat_ctx*& c : extra) if (c) { necat_ctx_destroy(c); c = nullptr; } }      // (a worker that leaves through _exit calls this itself)
    ~PmLanes() { close(); }
    PmLanes(const PmLanes&) = delete; PmLanes& operator=(const PmLanes&) = delete;
};

// `preloaded` (optional): the reference volume, already being read (oc2pmov reads it while the HIP runtime starts)
// `units` (optional): this process's share of the job, query volumes ascending; nullptr = the whole job (every query volume vid .. V-1)
inline int pm_run_volume(necat_ctx* ctx, const VolumesInfo& vi, int vid, const necat_map_options& opt, const char* output, const char* tag, const PmTrace& tr,
                         std::future<std::unique_ptr<PmLoaded>>* preloaded = nullptr, const std::vector<PmUnit>* units = nullptr, PmLanes* lanes = nullptr)
{
    auto fail = [&](const char* what, const char* detail) { fprintf(stderr, "[%s] ERROR: %s: %s\n", tag, what, detail); return 1; };
    std::vector<PmUnit> whole;
    if (!units) { for (int i = vid; i < vi.num_volumes; ++i) whole.push_back(PmUnit{i, 0, kPmSlots}); units = &whole; }
    // everything the job holds, released on EVERY way out (a resident oc2pm worker outlives a failed job)
    struct Held {
        necat_ctx* ctx; necat_volume* ref = nullptr; necat_index* ix = nullptr; FILE* out = nullptr;
        std::string tmp_out;
        ~Held()
        {
            if (out) { fclose(out); remove(tmp_out.c_str()); }
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
    // ---- the job's units (pm_worker.c:372-390), on L pair lanes ----
    // Lane l (a host thread; lane 0 on `ctx`, lane l >= 1 on a context of its own from `lanes`) takes the units l, l + L, l + 2 L, ...: it reads its NEXT unit's query
    // volume from disk while the current one is mapped, uploads, maps against the job's ONE index and reference volume (read-only: shared by every lane), and leaves
    // the records in the unit's slot.  This thread writes the slots IN UNIT ORDER - the job's file is byte for byte what one lane writes - while the lanes go on
    // (at most L + 1 finished units wait in memory).  L = 1 is the sequence of round 5 with the writing taken off the mapping thread.
    const size_t nu = units->size();
    const int L = (int)std::max<size_t>(1, std::min<size_t>(lanes ? (size_t)lanes->for_job(href.nbases, nu) : 1, nu));
    struct UnitOut {
        int state = 0;                   // 0 = not yet, 1 = records ready, 2 = failed (message printed), 3 = skipped after another unit's failure
        necat_m4* m4 = nullptr; uint64_t nm4 = 0; necat_candidate* cands = nullptr; uint64_t ncand = 0;
        uint32_t* recs = nullptr; uint64_t* poff = nullptr; int np = 0;          // NECAT_PM_PARTITIONS: the unit's share of the consensus partitions
        std::shared_ptr<PmLoaded> hq;    // the query volume on the host (read names of a -u 1 text output); null = the reference volume itself
        double secs = 0;
        void release() { necat_free(m4); necat_free(cands); necat_free(recs); necat_free(poff); m4 = nullptr; cands = nullptr; recs = nullptr; poff = nullptr; hq.reset(); }
    };
    std::vector<UnitOut> outs(nu);
    std::mutex mu; std::condition_variable cv;
    size_t written = 0; bool stop = false;
    auto job_name = [&](const PmUnit& u, char* job, size_t n) {
        if (u.slot_lo == 0 && u.slot_hi == kPmSlots) snprintf(job, n, "pairwise mapping v%d vs v%d", u.query_vol, vid);
        else snprintf(job, n, "pairwise mapping v%d vs v%d (query chunks %d..%d of %d)", u.query_vol, vid, u.slot_lo, u.slot_hi - 1, kPmSlots);
    };
    auto lane_main = [&](int l) {
        necat_ctx* c = ctx;
        bool lane_ok = true;
        if (l > 0) {
            necat_ctx*& slot = lanes->extra[(size_t)l - 1];
            if (!slot && necat_ctx_create(lanes->device, &slot)) { slot = nullptr; lane_ok = false; fail("pair lane", necat_last_error(nullptr)); }
            c = slot;
        }
        necat_volume* reads = nullptr; int cur_vol = -1;
        std::shared_ptr<PmLoaded> own;
        std::future<std::unique_ptr<PmLoaded>> next; int next_vol = -1;
        auto next_other = [&](size_t k) -> int {      // this lane's first unit after unit k whose query volume is neither unit k's nor the reference
            for (size_t j = k + (size_t)L; j < nu; j += (size_t)L) if ((*units)[j].query_vol != (*units)[k].query_vol && (*units)[j].query_vol != vid) return (*units)[j].query_vol;
            return -1;
        };
        auto prefetch = [&](int v) { if (v >= 0) { next = pm_load_async(vi, v); next_vol = v; } else next_vol = -1; };
        if ((size_t)l < nu && lane_ok) prefetch((*units)[(size_t)l].query_vol != vid ? (*units)[(size_t)l].query_vol : next_other((size_t)l));
        for (size_t k = (size_t)l; k < nu; k += (size_t)L) {
            UnitOut& o = outs[k];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || k < written + (size_t)L + 1; });
                if (stop || !lane_ok) { o.state = lane_ok ? 3 : 2; stop = true; cv.notify_all(); continue; }
            }
            const PmUnit& u = (*units)[k];
            const int i = u.query_vol;
            int st = 0;
            const double t1 = now_sec();
            if (i < vid || i >= vi.num_volumes || u.slot_lo < 0 || u.slot_hi > kPmSlots || u.slot_lo >= u.slot_hi) st = fail("unit", "outside the job");
            const bool whole_pair = u.slot_lo == 0 && u.slot_hi == kPmSlots;
            const int chunk_reads = st ? 0 : pair_chunk_reads((uint64_t)vi.read_count[i], kPmSlots);
            if (!st && i == vid) {
                if (reads && reads != ref) necat_volume_free(c, reads);
                reads = ref; cur_vol = vid;
            } else if (!st && i != cur_vol) {
                if (reads && reads != ref) { necat_volume_free(c, reads); }
                reads = nullptr; cur_vol = -1;
                if (next_vol != i) { if (next.valid()) next.wait(); prefetch(i); }
                own = std::shared_ptr<PmLoaded>(next.get().release());
                prefetch(next_other(k));
                if (!own->ok) st = fail("volume", own->err.c_str());
                else if (necat_volume_upload(c, own->v.pac.data(), own->v.nbases, own->v.offset.data(), own->v.size.data(), own->v.offset.size(), &reads))
                    { reads = nullptr; st = fail("necat_volume_upload", necat_last_error(c)); }
                else cur_vol = i;
            }
            if (!st) {
                if (i != vid) o.hq = own;
                const int read_start = vi.read_start_id[i];
                int rc2;
                if (opt.job == 1) {
                    // pm_search_one_volume with -j 1: seeding + extension, the candidates stay on the device
                    rc2 = whole_pair ? necat_map_pair(c, ix, ref, reads, read_start, ref_start, 1, &opt, 1 /* ONC_TAIL_MATCH_LEN_SHORT */, &o.m4, &o.nm4, &o.ncand)
                                     : necat_map_pair_part(c, ix, ref, reads, read_start, ref_start, 1, &opt, 1, chunk_reads, u.slot_lo, u.slot_hi, kPmSlots, &o.m4, &o.nm4, &o.ncand);
                    if (rc2) st = fail("necat_map_pair", necat_last_error(c));
                    else tr.stage("mapped", i, vid);
                } else {
                    rc2 = whole_pair ? necat_find_candidates(c, ix, ref, reads, read_start, ref_start, 1, &opt, &o.cands, &o.ncand)
                                     : necat_find_candidates_part(c, ix, ref, reads, read_start, ref_start, 1, &opt, chunk_reads, u.slot_lo, u.slot_hi, kPmSlots, &o.cands, &o.ncand);
                    if (rc2) st = fail("necat_find_candidates", necat_last_error(c));
                    else {
                        tr.stage("candidates found", i, vid);
                        // NECAT_PM_PARTITIONS=<batch size>: the consensus stage's partitions (oc2pcan's candidates.p<i>) straight from the
                        // candidates of this job, partitioned on the device (necat_pcan_partition) - the pipeline can then skip oc2pcan's
                        // write + read of the whole candidates file
                        if (pcan_batch > 0 && (rc2 = necat_pcan_partition(c, o.cands, o.ncand, pcan_batch, vi.num_reads, &o.recs, &o.poff, &o.np)))
                            st = fail("necat_pcan_partition", necat_last_error(c));
                    }
                }
            }
            o.secs = now_sec() - t1;
            std::lock_guard<std::mutex> lk(mu);
            o.state = st ? 2 : 1;
            if (st) stop = true;
            cv.notify_all();
        }
        if (next.valid()) next.wait();
        if (reads && reads != ref && c) necat_volume_free(c, reads);
    };
    std::vector<std::thread> lane_threads;
    for (int l = 0; l < L; ++l) lane_threads.emplace_back(lane_main, l);
    int status = 0;
    for (size_t k = 0; k < nu && !status; ++k) {
        UnitOut& o = outs[k];
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return o.state != 0; }); }
        if (o.state != 1) { status = 1; break; }
        const PmUnit& u = (*units)[k];
        const int i = u.query_vol, read_start = vi.read_start_id[i];
        const HostVolume* hreads = o.hq ? &o.hq->v : &href;
        char job[256];
        job_name(u, job, sizeof job);
        log_line("", job);
        bool wok = true;
        if (opt.job == 1) {
            necat_m4* const m4 = o.m4; const uint64_t nm4 = o.nm4;
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
        } else {
            const necat_candidate* const cands = o.cands; const uint64_t ncand = o.ncand;
            if (pcan_batch > 0) {
                for (int p = 0; p < o.np && wok; ++p) {
                    if (o.poff[p + 1] == o.poff[p]) continue;
                    FILE* pf = fopen((std::string(output) + ".p" + std::to_string(p)).c_str(), "ab");
                    wok = pf && fwrite(o.recs + 7 * o.poff[p], 28, o.poff[p + 1] - o.poff[p], pf) == o.poff[p + 1] - o.poff[p];
                    if (pf && fclose(pf) != 0) wok = false;
                }
                tr.stage("partitions written", i, vid);
            }
            if (opt.binary_output) {
                std::vector<uint32_t> items((size_t)ncand * 7);
                for (uint64_t k2 = 0; k2 < ncand; ++k2) pack_candidate(&cands[k2], items.data() + 7 * k2);
                wok = ncand == 0 || fwrite(items.data(), 28, ncand, out) == ncand;
            } else wok = write_records(out, ncand, 13 * 24, opt.num_threads, [&](char* p, uint64_t k2) { return put_candidate(p, cands[k2]); });     // DUMP_GAPPED_CANDIDATE (gapped_candidate.h:26-42)
        }
        o.release();
        if (!wok) status = fail("output", "write failed");
        else { tr.stage("records written", i, vid); log_line("[%s] INFO: '%s' takes %.2lf secs.\n", job, o.secs); }
        std::lock_guard<std::mutex> lk(mu);
        written = k + 1;
        cv.notify_all();
    }
    { std::lock_guard<std::mutex> lk(mu); if (status) stop = true; cv.notify_all(); }
    for (auto& t : lane_threads) t.join();
    for (auto& o : outs) o.release();
    FILE* f = H.out; H.out = nullptr;                 // closed here: the guard only cleans up after a failure
    if (fclose(f) != 0 && !status) status = fail("output", "write failed");
    if (!status && rename(H.tmp_out.c_str(), output) != 0) status = fail("output", "rename failed");
    if (status) remove(H.tmp_out.c_str());
    tr.stage("job done");
    return status;
}

}  // namespace necat_host
