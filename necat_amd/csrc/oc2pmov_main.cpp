// oc2pmov - drop-in replacement of NECAT's oc2pmov (pm_one_volume/main.c, pm_worker.c:338 pm_main):
//   oc2pmov [options] wrk-dir volume-id output
// Same argv, same volume inputs, same candidate / M4 record files; the work runs on one MI355X
// through libnecat_hip.so (device from NECAT_GPU, default 0).  There is no CPU fallback: without a
// usable GPU the program exits 1, like every other fatal error of the reference (OC_ERROR).
#include "host_io.h"
#include "host_fmt.h"

using namespace necat_host;

static int fail(const char* what, const char* detail)
{
    fprintf(stderr, "[oc2pmov] ERROR: %s: %s\n", what, detail);
    return 1;
}

// NECAT_CLI_TRACE=1: wall clock of the program's stages on stderr (where a cold start goes)
static double g_t0;
static void stage(const char* what)
{
    static const bool on = getenv("NECAT_CLI_TRACE") && atoi(getenv("NECAT_CLI_TRACE"));
    if (on) fprintf(stderr, "[oc2pmov] %8.1f ms  %s\n", (now_sec() - g_t0) * 1e3, what);
}

int main(int argc, char** argv)
{
    g_t0 = now_sec();
    necat_map_options opt;
    necat_default_options(&opt);
    if (argc < 4) {     // pm_one_volume/main.c:30-33
        fprintf(stderr, "USAGE:\n%s [options] wrk-dir volume-id output\n\nOPTIONS AND DESCRIPTIONS:\n", argv[0]);
        describe_options(stderr, &opt);
        return 1;
    }
    if (!parse_options(argc - 3, argv, &opt)) {
        fprintf(stderr, "USAGE:\n%s [options] wrk-dir volume-id output\n\nOPTIONS AND DESCRIPTIONS:\n", argv[0]);
        necat_map_options d; necat_default_options(&d);
        describe_options(stderr, &d);
        return 1;
    }
    const char* wrk_dir = argv[argc - 3];
    const int vid = atoi(argv[argc - 2]);
    const char* output = argv[argc - 1];

    std::string err;
    VolumesInfo vi;
    if (!load_volumes_info(wrk_dir, &vi, &err)) return fail("volume directory", err.c_str());
    if (vid < 0 || vid >= vi.num_volumes) return fail("volume id", "out of range");
    const char* dev_env = getenv("NECAT_GPU");
    // a fresh process pays for the VRAM the previous one dirtied (30 - 55 ms per GB on MI355X): keep the band-record pools small
    setenv("NECAT_BAND_POOL_MB", "4096", 0);
    necat_ctx* ctx = nullptr;
    int rc = necat_ctx_create(dev_env ? atoi(dev_env) : 0, &ctx);
    if (rc) return fail("GPU", "no usable gfx950 device (libnecat_hip has no CPU fallback)");
    stage("context created");

    HostVolume href;
    if (!load_volume(vi.names[vid].c_str(), &href, &err)) return fail("volume", err.c_str());
    stage("volume read");
    necat_volume* ref = nullptr;
    if ((rc = necat_volume_upload(ctx, href.pac.data(), href.nbases, href.offset.data(), href.size.data(), href.offset.size(), &ref)))
        return fail("necat_volume_upload", necat_last_error(ctx));
    stage("volume uploaded");
    log_line("", "build_lookup_table");
    double t0 = now_sec();
    necat_index* ix = nullptr;
    if ((rc = necat_index_build(ctx, ref, opt.kmer_size, opt.kmer_cnt_cutoff, &ix))) return fail("necat_index_build", necat_last_error(ctx));
    log_line("[%s] INFO: '%s' takes %.2lf secs.\n", "build_lookup_table", now_sec() - t0);

    stage("index built");
    // write to a temporary name first: a failed run never leaves a complete-looking pm_result_i
    const std::string tmp_out = std::string(output) + ".part";
    FILE* out = fopen(tmp_out.c_str(), "w");
    if (!out) return fail("output", "cannot open for writing");
    const int ref_start = vi.read_start_id[vid];
    uint64_t n_records = 0;
    for (int i = vid; i < vi.num_volumes; ++i) {     // pm_worker.c:372-390
        char job[256];
        snprintf(job, sizeof job, "pairwise mapping v%d vs v%d", i, vid);
        log_line("", job);
        t0 = now_sec();
        HostVolume hreads_own; const HostVolume* hreads = &href;
        necat_volume* reads = ref;
        if (i != vid) {
            if (!load_volume(vi.names[i].c_str(), &hreads_own, &err)) return fail("volume", err.c_str());
            hreads = &hreads_own;
            if ((rc = necat_volume_upload(ctx, hreads_own.pac.data(), hreads_own.nbases, hreads_own.offset.data(), hreads_own.size.data(),
                                          hreads_own.offset.size(), &reads))) return fail("necat_volume_upload", necat_last_error(ctx));
        }
        const int read_start = vi.read_start_id[i];
        necat_candidate* cands = nullptr; uint64_t ncand = 0;
        if (opt.job == 1) {
            // pm_search_one_volume with -j 1: seeding + extension, the candidates stay on the device
            necat_m4* m4 = nullptr; uint64_t nm4 = 0;
            if ((rc = necat_map_pair(ctx, ix, ref, reads, read_start, ref_start, 1, &opt, 1 /* ONC_TAIL_MATCH_LEN_SHORT */, &m4, &nm4, &ncand)))
                return fail("necat_map_pair", necat_last_error(ctx));
            stage("mapped");
            bool wok;
            if (opt.binary_output) wok = nm4 == 0 || fwrite(m4, sizeof(necat_m4), nm4, out) == nm4;
            else {
                const bool hdr = opt.use_hdr_as_id != 0;
                size_t max_len = 12 * 24;
                if (hdr) { size_t lq = 0, ls = 0; for (uint64_t r = 0; r < hreads->offset.size(); ++r) lq = std::max(lq, strlen(hreads->name(r)));
                           for (uint64_t r = 0; r < href.offset.size(); ++r) ls = std::max(ls, strlen(href.name(r))); max_len += lq + ls; }
                wok = write_records(out, nm4, max_len, opt.num_threads, [&](char* p, uint64_t k) {
                    const necat_m4& m = m4[k];
                    return hdr ? put_m4(p, m, hreads->name((uint64_t)(m.qid - read_start)), href.name((uint64_t)(m.sid - ref_start))) : put_m4(p, m, nullptr, nullptr);
                });
            }
            if (!wok) return fail("output", "write failed");
            n_records += nm4;
            necat_free(m4);
            stage("records written");
        } else {
            if ((rc = necat_find_candidates(ctx, ix, ref, reads, read_start, ref_start, 1, &opt, &cands, &ncand)))
                return fail("necat_find_candidates", necat_last_error(ctx));
            bool wok;
            if (opt.binary_output) {
                std::vector<uint32_t> items((size_t)ncand * 7);
                for (uint64_t k = 0; k < ncand; ++k) pack_candidate(&cands[k], items.data() + 7 * k);
                wok = ncand == 0 || fwrite(items.data(), 28, ncand, out) == ncand;
            } else wok = write_records(out, ncand, 13 * 24, opt.num_threads, [&](char* p, uint64_t k) { return put_candidate(p, cands[k]); });
            if (!wok) return fail("output", "write failed");
            n_records += ncand;
        }
        necat_free(cands);
        if (reads != ref) necat_volume_free(ctx, reads);
        log_line("[%s] INFO: '%s' takes %.2lf secs.\n", job, now_sec() - t0);
    }
    if (fclose(out) != 0) return fail("output", "write failed");
    if (rename(tmp_out.c_str(), output) != 0) return fail("output", "rename failed");
    stage("output closed");
    necat_index_free(ctx, ix);
    necat_volume_free(ctx, ref);
    necat_ctx_destroy(ctx);
    stage("context destroyed");
    return 0;
}
