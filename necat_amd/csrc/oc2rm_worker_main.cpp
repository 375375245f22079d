// oc2rm_worker - drop-in replacement of NECAT's oc2rm_worker (reference_mapping/rm_one_vol_main.c; necat.pl:661 runs one per volume):
//   oc2rm_worker [options] wrk-dir reference output [-mn node_id num_nodes]
// Maps the reads of the volumes node_id, node_id + num_nodes, ... of wrk-dir against `reference` (one volume file, as oc2mkdb
// writes them) and writes the M4 records (DUMP_ASM_M4 / DUMP_ASM_M4_HDR_ID, m4_record.h:72-124; 96-byte records with -u 1).
// Options and defaults: sDefaultReferenceMapingOptions (common/map_options.c:31-46).  The lookup table, the candidates and the
// block-wise alignments run on one MI355X through libnecat_hip.so (necat_map_reference; device from NECAT_GPU, default 0), the
// reference's rescue pair for alignments that fall short on the host threads.  Records come out in read order (the reference's
// order with -t 1; with more threads its order depends on scheduling).  There is no CPU fallback: without a usable GPU the
// program exits 1.
#include "pm_job.h"

using namespace necat_host;

static int fail(const char* what, const char* detail)
{
    fprintf(stderr, "[oc2rm_worker] ERROR: %s: %s\n", what, detail);
    return 1;
}

static void rm_defaults(necat_map_options* o)
{   // map_options.c:31-46
    necat_default_options(o);
    o->kmer_size = 15; o->scan_window = 5; o->kmer_cnt_cutoff = 500; o->block_size = 1000; o->block_score_cutoff = 3;
    o->num_candidates = 20; o->align_size_cutoff = 400; o->ddfs_cutoff = 0.25; o->error = 0.5; o->num_output = 20;
    o->num_threads = 1; o->job = 1; o->binary_output = 0; o->use_hdr_as_id = 1;
}

static int usage(const char* prog)
{   // rm_one_vol_main.c:9-21
    necat_map_options d;
    necat_default_options(&d);              // the reference describes the pairwise defaults here (:20)
    fprintf(stderr, "USAGE:\n%s [OPTIONS] wrk-dir reference output\n\nIf Multiple Nodes Are Used:\n%s [OPTIONS] wrk_dir reference output -mn node_id num_nodes\n\n"
                    "OPTIONS AND DESCRIPTIONS:\n", prog, prog);
    describe_options(stderr, &d);
    return 1;
}

int main(int argc, char** argv)
{
    necat_host::necat_cli_env();          // (before the first HIP call: host_io.h)
    const PmTrace tr;
    int svid = 0, num_nodes = 1;
    if (argc >= 7 && strcmp(argv[argc - 3], "-mn") == 0) { svid = atoi(argv[argc - 2]); num_nodes = atoi(argv[argc - 1]); argc -= 3; }
    necat_map_options opt;
    rm_defaults(&opt);
    if (argc < 4 || !parse_options(argc - 3, argv, &opt) || num_nodes < 1 || svid < 0) return usage(argv[0]);
    const char* wrk_dir = argv[argc - 3];
    const char* reference_path = argv[argc - 2];
    const char* output = argv[argc - 1];

    std::string err;
    VolumesInfo vi;
    if (!load_volumes_info(wrk_dir, &vi, &err)) return fail("volume directory", err.c_str());
    auto ref_l = std::async(std::launch::async, [&]() { auto l = std::make_unique<PmLoaded>(); l->ok = load_volume(reference_path, &l->v, &l->err); return l; });
    const char* dev_env = getenv("NECAT_GPU");
    setenv("NECAT_BAND_POOL_MB", "1024", 0);
    necat_ctx* ctx = nullptr;
    if (necat_ctx_create(dev_env ? atoi(dev_env) : 0, &ctx)) { ref_l.wait(); return fail("GPU", "no usable gfx950 device (libnecat_hip has no CPU fallback)"); }
    tr.stage("context created");
    std::unique_ptr<PmLoaded> href_l = ref_l.get();
    if (!href_l->ok) { necat_ctx_destroy(ctx); return fail("reference", href_l->err.c_str()); }
    const HostVolume& href = href_l->v;
    int status = 0;
    necat_volume* ref = nullptr;
    necat_index* ix = nullptr;
    FILE* out = nullptr;
    const std::string tmp_out = std::string(output) + ".part";
    do {
        if (necat_volume_upload(ctx, href.pac.data(), href.nbases, href.offset.data(), href.size.data(), href.offset.size(), &ref)) { status = fail("necat_volume_upload", necat_last_error(ctx)); break; }
        log_line("", "build_lookup_table");
        double t0 = now_sec();
        if (necat_index_build(ctx, ref, opt.kmer_size, opt.kmer_cnt_cutoff, &ix)) { status = fail("necat_index_build", necat_last_error(ctx)); break; }
        log_line("[%s] INFO: '%s' takes %.2lf secs.\n", "build_lookup_table", now_sec() - t0);
        tr.stage("index built");
        out = fopen(tmp_out.c_str(), "w");
        if (!out) { status = fail("output", "cannot open for writing"); break; }
        std::future<std::unique_ptr<PmLoaded>> next;
        if (svid < vi.num_volumes) next = pm_load_async(vi, svid);
        for (int i = svid; i < vi.num_volumes && !status; i += num_nodes) {          // rm_one_vol_main.c:63-88
            char job[128];
            snprintf(job, sizeof job, "mapping volume %d", i);
            log_line("", job);
            t0 = now_sec();
            std::unique_ptr<PmLoaded> vol = next.get();
            if (i + num_nodes < vi.num_volumes) next = pm_load_async(vi, i + num_nodes);
            if (!vol->ok) { status = fail("volume", vol->err.c_str()); break; }
            const HostVolume& hreads = vol->v;
            necat_volume* reads = nullptr;
            if (necat_volume_upload(ctx, hreads.pac.data(), hreads.nbases, hreads.offset.data(), hreads.size.data(), hreads.offset.size(), &reads)) { status = fail("necat_volume_upload", necat_last_error(ctx)); break; }
            const int read_start = vi.read_start_id[i];
            necat_m4* m4 = nullptr; uint64_t nm4 = 0, ncand = 0, nresc = 0;
            if (necat_map_reference(ctx, ix, ref, reads, read_start, 0, &opt, &m4, &nm4, &ncand, &nresc)) status = fail("necat_map_reference", necat_last_error(ctx));
            else {
                tr.stage("mapped", i, 0);
                bool wok;
                if (opt.binary_output) wok = nm4 == 0 || fwrite(m4, sizeof(necat_m4), nm4, out) == nm4;
                else {
                    const bool hdr = opt.use_hdr_as_id != 0;
                    size_t max_len = 12 * 24;
                    if (hdr) {
                        size_t lq = 0, ls = 0;
                        for (uint64_t r = 0; r < hreads.offset.size(); ++r) lq = std::max(lq, strlen(hreads.name(r)));
                        for (uint64_t r = 0; r < href.offset.size(); ++r) ls = std::max(ls, strlen(href.name(r)));
                        max_len += lq + ls;
                    }
                    wok = write_records(out, nm4, max_len, opt.num_threads, [&](char* p, uint64_t k) {
                        const necat_m4& m = m4[k];
                        return hdr ? put_m4(p, m, hreads.name((uint64_t)(m.qid - read_start)), href.name((uint64_t)m.sid)) : put_m4(p, m, nullptr, nullptr);
                    });
                }
                necat_free(m4);
                if (!wok) status = fail("output", "write failed");
            }
            necat_volume_free(ctx, reads);
            if (!status) log_line("[%s] INFO: '%s' takes %.2lf secs.\n", job, now_sec() - t0);
        }
        if (next.valid()) next.wait();
    } while (0);
    if (out && fclose(out) != 0 && !status) status = fail("output", "write failed");
    if (out && !status && rename(tmp_out.c_str(), output) != 0) status = fail("output", "rename failed");
    if (status) remove(tmp_out.c_str());
    necat_index_free(ctx, ix);
    necat_volume_free(ctx, ref);
    necat_ctx_destroy(ctx);
    tr.stage("done");
    return status;
}
