// asm_job.h - one oc2asmpm job (asm_pm/asmpm.c:11-68): reference volume `vid` against the volumes vid .. V-1.  Per volume pair: the block vote and the
// chained range of every read on the device (necat_asm_plan_batch, asm_plan.h; 4 % + 20 % of the reference's time), ALL anchors of the pair in one
// call of the device's block aligner (necat_asm_align_batch; 74 %), then DALIGNER's end extension and the records on the host threads
// (asm_core.h Extender / BatchMapper::finish).
#pragma once
#include <atomic>
#include <memory>
#include <thread>

#include "asm_core.h"
#include "host_fmt.h"
#include "host_io.h"

namespace necat_host {

struct HostCodes {       // a volume's bases, one code per byte
    std::vector<uint8_t> c;
    const HostVolume* v = nullptr;
    void set(const HostVolume& hv)
    {
        v = &hv; c.resize(hv.nbases);
        for (uint64_t i = 0; i < hv.nbases; ++i) c[i] = (uint8_t)((hv.pac[i >> 2] >> ((~i & 3) << 1)) & 3);
    }
    void strand(uint64_t id, int rev, std::vector<uint8_t>& out) const
    {
        const uint64_t b = v->offset[id], n = v->size[id];
        out.resize(n);
        if (!rev) memcpy(out.data(), c.data() + b, n);
        else for (uint64_t i = 0; i < n; ++i) out[i] = (uint8_t)(3 - c[b + n - 1 - i]);
    }
};

template <class F>
inline void asm_parallel(uint64_t n, int threads, F&& fn)
{
    unsigned nt = (unsigned)std::max(1, threads);
    nt = (unsigned)std::min<uint64_t>(nt, std::max<uint64_t>(1, n));
    std::atomic<uint64_t> next(0);
    auto work = [&](unsigned tid) { for (;;) { const uint64_t b = next.fetch_add(8); if (b >= n) break; for (uint64_t i = b; i < std::min(n, b + 8); ++i) fn(i, tid); } };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
}

// returns 0, or 1 after printing "[tag] ERROR: ..."
inline int asm_run_volume(necat_ctx* ctx, const VolumesInfo& vi, int vid, const necat_map_options& opt, const char* output, const char* tag)
{
    using namespace necat;
    auto fail = [&](const char* what, const char* detail) { fprintf(stderr, "[%s] ERROR: %s: %s\n", tag, what, detail); return 1; };
    std::string err;
    HostVolume href;
    const bool cli_trace = getenv("NECAT_CLI_TRACE") != nullptr;
    const double t_job = now_sec();
    if (!load_volume(vi.names[vid].c_str(), &href, &err)) return fail("volume", err.c_str());
    const double t_loaded = now_sec();
    // everything the job holds on the device / in files, released on every way out (the context may outlive a failed job)
    struct Held {
        necat_ctx* ctx; necat_volume* ref = nullptr; necat_volume* reads = nullptr; necat_index* ix = nullptr; FILE* out = nullptr; std::string tmp_out;
        ~Held()
        {
            if (out) { fclose(out); remove(tmp_out.c_str()); }
            if (ix) necat_index_free(ctx, ix);
            if (reads && reads != ref) necat_volume_free(ctx, reads);
            if (ref) necat_volume_free(ctx, ref);
        }
    } H{ctx};
    if (necat_volume_upload(ctx, href.pac.data(), href.nbases, href.offset.data(), href.size.data(), href.offset.size(), &H.ref)) return fail("necat_volume_upload", necat_last_error(ctx));
    necat_volume* const ref = H.ref;
    log_line("", "build_lookup_table");
    double t0 = now_sec();
    if (necat_index_build(ctx, ref, opt.kmer_size, opt.kmer_cnt_cutoff, &H.ix)) return fail("necat_index_build", necat_last_error(ctx));
    log_line("[%s] INFO: '%s' takes %.2lf secs.\n", "build_lookup_table", now_sec() - t0);
    const double t_indexed = now_sec();
    HostCodes cref; cref.set(href);
    if (cli_trace) fprintf(stderr, "[oc2asmpm] volume %d: read %.2f s, upload + index %.2f s, one-byte codes %.2f s\n", vid, t_loaded - t_job, t_indexed - t_loaded, now_sec() - t_indexed);
    auto subject_of = [&](int sid, int strand, std::vector<uint8_t>& s) { cref.strand((uint64_t)sid, strand, s); };

    H.tmp_out = std::string(output) + ".part";
    const std::string& tmp_out = H.tmp_out;
    H.out = fopen(tmp_out.c_str(), "w");
    if (!H.out) return fail("output", "cannot open for writing");
    FILE* const out = H.out;
    const int ref_start = vi.read_start_id[vid];
    const int nthreads = std::max(1, std::min(opt.num_threads, 256));
    int status = 0;
    for (int i = vid; i < vi.num_volumes && !status; ++i) {         // asmpm.c:38-58
        char job[256];
        snprintf(job, sizeof job, "pairwise mapping v%d vs v%d", vid, i);
        log_line("", job);
        t0 = now_sec();
        HostVolume own;
        const HostVolume* hreads = &href;
        H.reads = ref;
        HostCodes cown;
        const HostCodes* crd = &cref;
        if (i != vid) {
            if (!load_volume(vi.names[i].c_str(), &own, &err)) { status = fail("volume", err.c_str()); break; }
            hreads = &own;
            H.reads = nullptr;
            if (necat_volume_upload(ctx, own.pac.data(), own.nbases, own.offset.data(), own.size.data(), own.offset.size(), &H.reads)) { H.reads = nullptr; status = fail("necat_volume_upload", necat_last_error(ctx)); break; }
            cown.set(own); crd = &cown;
        }
        necat_volume* const reads = H.reads;
        const int read_start = vi.read_start_id[i];
        const uint64_t nreads = hreads->offset.size();
        // phase A: votes and chained ranges of every read of the volume, on the device (asm_plan.h)
        std::vector<std::vector<asmpm::Planned>> planned(nreads);
        {
            necat_asm_plan* plans = nullptr; uint64_t* first = nullptr;
            if (necat_asm_plan_batch(ctx, H.ix, ref, reads, read_start, ref_start, &opt, &plans, &first)) { status = fail("necat_asm_plan_batch", necat_last_error(ctx)); break; }
            for (uint64_t r = 0; r < nreads; ++r) {
                planned[r].resize((size_t)(first[r + 1] - first[r]));
                for (uint64_t q = first[r]; q < first[r + 1]; ++q) {
                    asmpm::Planned& p = planned[r][(size_t)(q - first[r])];
                    p.sid = plans[q].sid - ref_start; p.sdir = plans[q].sdir; p.qoff = plans[q].qoff; p.soff = plans[q].soff; p.score = plans[q].score; p.ssize = plans[q].ssize;
                }
            }
            necat_free(plans); necat_free(first);
        }
        const double t_plan = now_sec() - t0;
        double t_dev = 0, t_fin = 0;
        // phases B and C over runs of reads whose anchors fit one call (the aligner keeps every anchor's columns: read + subject bytes each)
        std::vector<std::vector<necat_m4>> recs(nreads);
        std::vector<asmpm::BatchMapper> mappers((size_t)nthreads);
        const uint64_t kMaxCallBytes = 2ULL << 30;
        const uint64_t kMaxCallAnchors = getenv("NECAT_ASM_CALL_ANCHORS") ? (uint64_t)std::max(1, atoi(getenv("NECAT_ASM_CALL_ANCHORS"))) : 1ULL << 20;      // (tests: several calls per pair)
        for (uint64_t r0 = 0; r0 < nreads && !status;) {
            // phase B: the anchors of reads [r0, r1) through the device's block aligner
            std::vector<uint64_t> first;
            std::vector<necat_asm_anchor> anchors;
            std::vector<int64_t> slot;                 // per planned candidate: its anchor or -1
            uint64_t r1 = r0, bytes = 0;
            for (; r1 < nreads; ++r1) {
                uint64_t add = 0, cnt = 0;
                for (const asmpm::Planned& p : planned[r1]) if (p.qoff >= 0) { add += hreads->size[r1] + (uint64_t)p.ssize + 64; ++cnt; }
                if (r1 > r0 && (bytes + add > kMaxCallBytes || anchors.size() + cnt > kMaxCallAnchors)) break;
                bytes += add;
                first.push_back(slot.size());
                for (const asmpm::Planned& p : planned[r1]) {
                    if (p.qoff < 0) { slot.push_back(-1); continue; }
                    necat_asm_anchor a;
                    a.qid = (int)r1 + read_start; a.sid = p.sid + ref_start; a.sdir = p.sdir; a.qoff = p.qoff; a.soff = p.soff;
                    slot.push_back((int64_t)anchors.size());
                    anchors.push_back(a);
                }
            }
            necat_alignment* aln = nullptr; uint8_t* cols = nullptr; uint64_t* cols_off = nullptr;
            const double tb = now_sec();
            if (necat_asm_align_batch(ctx, ref, reads, read_start, ref_start, anchors.data(), anchors.size(), 0.5 /* hbn_align.c:8 */, 400 /* asm_pm_common.c:354 */,
                                      &aln, &cols, &cols_off)) { status = fail("necat_asm_align_batch", necat_last_error(ctx)); }
            const double tc = now_sec();
            t_dev += tc - tb;
            // phase C: end extension, records
            if (!status) {
                asm_parallel(r1 - r0, nthreads, [&](uint64_t k0, unsigned tid) {
                    const uint64_t r = r0 + k0;
                    const size_t n = planned[r].size();
                    if (n == 0) return;
                    std::vector<uint8_t> fwd;
                    crd->strand(r, 0, fwd);
                    std::vector<asmpm::BlockAlignment> ba(n);
                    std::unique_ptr<bool[]> ok(new bool[n]);
                    std::vector<const uint8_t*> ops(n, nullptr);
                    std::vector<size_t> ncols(n, 0);
                    for (size_t k = 0; k < n; ++k) {
                        ok[k] = false;
                        const int64_t s = slot[first[k0] + k];
                        if (s < 0) continue;
                        const necat_alignment& a = aln[s];
                        ok[k] = a.ok != 0;
                        if (!ok[k]) continue;
                        ba[k].qoff = a.qoff; ba[k].qend = a.qend; ba[k].toff = a.toff; ba[k].tend = a.tend; ba[k].ident_perc = a.ident_perc;
                        ops[k] = cols + cols_off[s]; ncols[k] = (size_t)a.align_size;
                    }
                    // (the alignment columns stay packed: only their two ends are looked at, asm_core.h Extender::ends_packed)
                    mappers[tid].finish_packed(planned[r].data(), n, ok.get(), ba.data(), ops.data(), ncols.data(), fwd.data(), (int)r + read_start, (int)fwd.size(), subject_of, recs[r]);
                    for (necat_m4& m : recs[r]) m.sid += ref_start;
                });
            }
            t_fin += now_sec() - tc;
            necat_free(aln); necat_free(cols); necat_free(cols_off);
            r0 = r1;
        }
        const double t_w0 = now_sec();
        if (!status) {
            std::vector<necat_m4> all;
            for (auto& v : recs) all.insert(all.end(), v.begin(), v.end());
            bool wok;
            if (opt.binary_output) {        // asm_pm_common.c:66-73: ids + 1
                for (necat_m4& m : all) { ++m.qid; ++m.sid; }
                wok = all.empty() || fwrite(all.data(), sizeof(necat_m4), all.size(), out) == all.size();
            } else {                        // DUMP_ASM_M4_HDR_ID (:74-83)
                size_t lq = 0, ls = 0;
                for (uint64_t r = 0; r < nreads; ++r) lq = std::max(lq, strlen(hreads->name(r)));
                for (uint64_t r = 0; r < href.offset.size(); ++r) ls = std::max(ls, strlen(href.name(r)));
                wok = write_records(out, all.size(), 12 * 24 + lq + ls, opt.num_threads, [&](char* p, uint64_t k) {
                    const necat_m4& m = all[k];
                    return put_m4(p, m, hreads->name((uint64_t)(m.qid - read_start)), href.name((uint64_t)(m.sid - ref_start)));
                });
            }
            if (!wok) status = fail("output", "write failed");
        }
        if (reads != ref) necat_volume_free(ctx, reads);
        H.reads = nullptr;
        if (cli_trace) fprintf(stderr, "[oc2asmpm] %s: votes + ranges (device) %.2f s, block aligner calls %.2f s, end extension + records %.2f s (%d host threads), output %.2f s\n", job, t_plan,
                               t_dev, t_fin, nthreads, now_sec() - t_w0);
        if (!status) log_line("[%s] INFO: '%s' takes %.2lf secs.\n", job, now_sec() - t0);
    }
    H.out = nullptr;                  // closed here: the guard only cleans up after an early return
    if (fclose(out) != 0 && !status) status = fail("output", "write failed");
    if (!status && rename(tmp_out.c_str(), output) != 0) status = fail("output", "rename failed");
    if (status) remove(tmp_out.c_str());
    return status;
}

}  // namespace necat_host
