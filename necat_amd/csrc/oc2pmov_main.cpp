// oc2pmov - drop-in replacement of NECAT's oc2pmov (pm_one_volume/main.c, pm_worker.c:338 pm_main):
//   oc2pmov [options] wrk-dir volume-id output
// Same argv, same volume inputs, same candidate / M4 record files; the work runs on one MI355X
// through libnecat_hip.so (device from NECAT_GPU, default 0).  There is no CPU fallback: without a
// usable GPU the program exits 1, like every other fatal error of the reference (OC_ERROR).
#include "pm_job.h"

using namespace necat_host;

static int fail(const char* what, const char* detail)
{
    fprintf(stderr, "[oc2pmov] ERROR: %s: %s\n", what, detail);
    return 1;
}

int main(int argc, char** argv)
{
    necat_host::necat_cli_env();          // (before the first HIP call: host_io.h)
    const PmTrace tr;
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
    setenv("NECAT_BAND_POOL_MB", "1024", 0);
    auto ref_volume = pm_load_async(vi, vid);        // read while the HIP runtime starts (0.1 - 0.2 s)
    necat_ctx* ctx = nullptr;
    int rc = necat_ctx_create(dev_env ? atoi(dev_env) : 0, &ctx);
    if (rc) { ref_volume.wait(); return fail("GPU", "no usable gfx950 device (libnecat_hip has no CPU fallback)"); }
    tr.stage("context created");
    // the job's query volumes on NECAT_PAIR_LANES lanes (default 2): a second context is only made when the job has a second unit, by its lane's thread, beside unit 0
    int status;
    {
        PmLanes lanes(dev_env ? atoi(dev_env) : 0);
        status = pm_run_volume(ctx, vi, vid, opt, output, "oc2pmov", tr, &ref_volume, nullptr, &lanes);
    }
    // (Round 6 tried leaving with _exit() right here - the output is closed and renamed, the driver reclaims the arenas with the process - to save the 25 - 30 ms of
    // necat_ctx_destroy: the NEXT process then waited ~ 450 ms in its first big allocation while the driver scrubbed what this one had left mapped (tools/r06/run4.sh:
    // 0.93 - 1.0 s per run against 0.49 - 0.6 s).  Memory handed back with hipFree is clean when the next process asks for it; so the context is taken apart in order.)
    necat_ctx_destroy(ctx);
    tr.stage("context destroyed");
    return status;
}
