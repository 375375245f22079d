// oc2asmpm - drop-in replacement of NECAT's oc2asmpm (asm_pm/asmpm.c; necat.pl:573,880,1000,1152: the overlapper of corrected reads):
//   oc2asmpm [map options] wrk_dir volume_id output
// Same argv (pairwise defaults with -n / -m 100, asmpm.c:13-14), same volume inputs, same M4 records: text with names (-u 0) or 96-byte records with
// ids + 1 (-u 1).  The lookup table and the block aligner (2048-bp blocks: three quarters of the reference's time) run on one MI355X through
// libnecat_hip.so (device from NECAT_GPU, default 0); the vote, the chained ranges and the end extension on -t host threads (asm_core.h).  Records
// come out in read order (the reference's order with -t 1).  There is no CPU fallback for the alignments: without a usable GPU the program exits 1.
#include "asm_job.h"

using namespace necat_host;

int main(int argc, char** argv)
{
    necat_host::necat_cli_env();          // (before the first HIP call: host_io.h)
    necat_map_options opt;
    necat_default_options(&opt);
    opt.num_candidates = opt.num_output = 100;           // MAXC, asmpm.c:14
    if (argc < 4 || !parse_options(argc - 3, argv, &opt)) {
        fprintf(stdout, "USAGE:\n%s [map options] wrk_dir volume_id output\n", argv[0]);      // asmpm.c:3-9
        return 1;
    }
    const char* wrk_dir = argv[argc - 3];
    const int vid = atoi(argv[argc - 2]);
    const char* output = argv[argc - 1];
    std::string err;
    VolumesInfo vi;
    if (!load_volumes_info(wrk_dir, &vi, &err)) { fprintf(stderr, "[oc2asmpm] ERROR: volume directory: %s\n", err.c_str()); return 1; }
    if (vid < 0 || vid >= vi.num_volumes) { fprintf(stderr, "[oc2asmpm] ERROR: volume id: out of range\n"); return 1; }
    const char* dev_env = getenv("NECAT_GPU");
    setenv("NECAT_BAND_POOL_MB", "32768", 0);            // 126 MB of band records per wave of 64 alignments
    necat_ctx* ctx = nullptr;
    if (necat_ctx_create(dev_env ? atoi(dev_env) : 0, &ctx)) { fprintf(stderr, "[oc2asmpm] ERROR: GPU: no usable gfx950 device (libnecat_hip has no CPU fallback)\n"); return 1; }
    const int status = asm_run_volume(ctx, vi, vid, opt, output, "oc2asmpm");
    necat_ctx_destroy(ctx);
    return status;
}
