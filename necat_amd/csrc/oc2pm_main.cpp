// oc2pm - drop-in replacement of NECAT's oc2pm wrapper (pairwise_mapping/main.c:79-118):
//   oc2pm [options] wrk-dir output
// For every volume not yet marked wrk-dir/pm<i>.finished, run `oc2pmov <options> wrk-dir i
// wrk-dir/pm_result_i`, then concatenate the per-volume results in volume order into `output` and
// delete them.  The child is looked up next to this binary first, then on PATH (as the reference does).
#include <libgen.h>
#include <limits.h>
#include <unistd.h>
#include <sys/wait.h>

#include "host_io.h"

using namespace necat_host;

int main(int argc, char** argv)
{
    necat_map_options opt;
    // sDefaultPairwiseMapingOptions (map_options.c:12-28); oc2pm applies them (main.c:83)
    opt.kmer_size = 15; opt.scan_window = 10; opt.kmer_cnt_cutoff = 500; opt.block_size = 2000; opt.block_score_cutoff = 3;
    opt.num_candidates = 500; opt.align_size_cutoff = 500; opt.ddfs_cutoff = 0.25; opt.error = 0.5; opt.num_output = 500;
    opt.num_threads = 1; opt.job = 1; opt.binary_output = 0; opt.use_hdr_as_id = 1;
    if (argc < 3 || !parse_options(argc - 2, argv, &opt)) {
        fprintf(stderr, "USAGE:\n%s [options] wrk-dir output\n\nOPTIONS AND DESCRIPTIONS:\n", argv[0]);
        describe_options(stderr, &opt);
        return 1;
    }
    const char* wrk_dir = argv[argc - 2];
    const char* output = argv[argc - 1];
    std::string err;
    VolumesInfo vi;
    if (!load_volumes_info(wrk_dir, &vi, &err)) { fprintf(stderr, "[oc2pm] ERROR: %s\n", err.c_str()); return 1; }
    char self[PATH_MAX];
    std::string child = "oc2pmov";
    ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    if (n > 0) { self[n] = 0; std::string cand = std::string(dirname(self)) + "/oc2pmov"; if (access(cand.c_str(), X_OK) == 0) child = cand; }
    const std::string base = dir_prefix(wrk_dir);
    // NECAT_GPUS=0,1,2,3 (default: the single device NECAT_GPU or 0): reference volumes are independent
    // jobs (necat.pl:190-202 sends them to grid nodes), so up to one child per GPU runs at a time,
    // heaviest volumes first (volume i is mapped against V - i volumes).
    std::vector<int> gpus;
    if (const char* e = getenv("NECAT_GPUS")) {
        for (const char* p = e; *p;) { gpus.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
    }
    if (gpus.empty()) gpus.push_back(getenv("NECAT_GPU") ? atoi(getenv("NECAT_GPU")) : 0);
    std::vector<int> todo;
    for (int i = 0; i < vi.num_volumes; ++i) {
        char fin[4096];
        snprintf(fin, sizeof fin, "%s/pm%d.finished", wrk_dir, i);            // main.c:55-70
        if (access(fin, F_OK) != 0) todo.push_back(i);
    }
    std::vector<pid_t> running(gpus.size(), 0);
    std::vector<int> running_vol(gpus.size(), -1);
    size_t next = 0, active = 0;
    bool failed = false;
    auto reap = [&](pid_t pid, int status) {
        for (size_t g = 0; g < gpus.size(); ++g) if (running[g] == pid) {
            const int v = running_vol[g];
            running[g] = 0; running_vol[g] = -1; --active;
            if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { fprintf(stderr, "[oc2pm] ERROR: oc2pmov for volume %d failed\n", v); failed = true; }
            else { char fin[4096]; snprintf(fin, sizeof fin, "%s/pm%d.finished", wrk_dir, v); FILE* f = fopen(fin, "w"); if (f) fclose(f); }
        }
    };
    while ((next < todo.size() && !failed) || active) {
        bool launched = false;
        for (size_t g = 0; g < gpus.size() && next < todo.size() && !failed; ++g) {
            if (running[g]) continue;
            const int v = todo[next++];
            char cmd[8192];
            snprintf(cmd, sizeof cmd, "NECAT_GPU=%d %s %s %s %d %spm_result_%d", gpus[g], child.c_str(), options_to_string(&opt).c_str(), wrk_dir, v, base.c_str(), v);
            fprintf(stdout, "Running command '%s'\n", cmd);
            fflush(stdout);
            const pid_t pid = fork();
            if (pid < 0) { fprintf(stderr, "[oc2pm] ERROR: fork failed\n"); return 1; }
            if (pid == 0) { execl("/bin/sh", "sh", "-c", cmd, (char*)nullptr); _exit(127); }
            running[g] = pid; running_vol[g] = v; ++active; launched = true;
        }
        if (active && !(launched && next < todo.size() && active < gpus.size())) {
            int status = 0;
            const pid_t pid = wait(&status);
            if (pid > 0) reap(pid, status);
        }
    }
    if (failed) return 1;
    FILE* out = fopen(output, "w");
    if (!out) { fprintf(stderr, "[oc2pm] ERROR: cannot open %s\n", output); return 1; }
    std::vector<char> buf(1 << 20);
    for (int i = 0; i < vi.num_volumes; ++i) {
        char res[4096];
        snprintf(res, sizeof res, "%spm_result_%d", base.c_str(), i);
        FILE* in = fopen(res, "r");
        if (!in) { fprintf(stderr, "[oc2pm] ERROR: missing %s\n", res); fclose(out); return 1; }
        size_t k;
        while ((k = fread(buf.data(), 1, buf.size(), in)) > 0) fwrite(buf.data(), 1, k, out);
        fclose(in);
        remove(res);
    }
    fclose(out);
    return 0;
}
