// oc2pm - drop-in replacement of NECAT's oc2pm wrapper (pairwise_mapping/main.c:79-118):
//   oc2pm [options] wrk-dir output
// For every volume not yet marked wrk-dir/pm<i>.finished, do what `oc2pmov <options> wrk-dir i wrk-dir/pm_result_i` does, then
// concatenate the per-volume results in volume order into `output` and delete them.  The reference forks one oc2pmov per volume;
// here ONE resident worker process per GPU (NECAT_GPUS=0,1,2,3; default the single device NECAT_GPU or 0) takes volumes from a
// shared counter and runs their jobs on one context (pm_job.h): HIP start-up, stream creation and pool allocation are paid once
// per GPU, not once per volume, and the next volume is read from disk while the current one is mapped.
// With several GPUs the unit of work is not the volume but the (reference volume, query volume) PAIR (NECAT_PM_SCHEDULE=pairs, the
// default for more than one GPU; =volumes keeps whole volumes from a shared counter): the pairs of all unfinished volumes are laid on
// one cost line and every worker takes an equal stretch of it, pairs at a boundary split by query reads (pair_sched.h) - the jobs
// of a project are very unequal (volume 0 has V query volumes, volume V - 1 one, the last volume is a remainder), and whole volumes
// would leave most GPUs idle while the first ones work.  Every worker writes its share of job v to pm_result_v.r<g>; the shares
// are concatenated to pm_result_v (then pm<v>.finished) - the same records the job run whole would have written.
#include <limits.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/wait.h>

#include "pm_job.h"
#include "pair_sched.h"

using namespace necat_host;

int main(int argc, char** argv)
{
    necat_host::necat_cli_env();          // (before the first HIP call: host_io.h)
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
    const std::string base = dir_prefix(wrk_dir);
    // reference volumes are independent jobs (necat.pl:190-202 sends them to grid nodes); heaviest first (volume i is mapped
    // against V - i volumes)
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
    bool failed = false;
    const char* sched = getenv("NECAT_PM_SCHEDULE");
    const bool pairs = !todo.empty() && (sched ? !strcmp(sched, "pairs") : gpus.size() > 1);
    std::vector<char> buf(1 << 20);
    // append the file `src` to `out`; a missing file is an error unless `optional`
    auto append_file = [&](FILE* out, const std::string& src, bool optional) -> bool {
        FILE* in = fopen(src.c_str(), "rb");
        if (!in) return optional;
        size_t k; bool ok = true;
        while (ok && (k = fread(buf.data(), 1, buf.size(), in)) > 0) ok = fwrite(buf.data(), 1, k, out) == k;
        if (ferror(in)) ok = false;
        fclose(in);
        return ok;
    };
    if (pairs) {
        const int V = vi.num_volumes, G = (int)gpus.size();
        std::vector<uint64_t> bases((size_t)V, 0);
        std::vector<uint8_t> skip((size_t)V, 1);
        for (int v : todo) skip[(size_t)v] = 0;
        for (int v = 0; v < V; ++v) if (!volume_bases(vi.names[(size_t)v].c_str(), &bases[(size_t)v], &err)) { fprintf(stderr, "[oc2pm] ERROR: %s\n", err.c_str()); return 1; }
        const PairSchedule S = pair_schedule(bases.data(), V, G, kPmSlots, skip.data());
        const int pcan_batch = (opt.job == 0 && getenv("NECAT_PM_PARTITIONS")) ? atoi(getenv("NECAT_PM_PARTITIONS")) : 0;
        const int np = pcan_batch > 0 ? (vi.num_reads + pcan_batch - 1) / pcan_batch : 0;
        // leftovers of an earlier, aborted run (another schedule, another batch size) must not be merged into this one's files: the shares
        // pm_result_<v>.r<g>[.p<p>] of every worker and the partition files pm_result_<v>.p<p> of the volumes still to do
        for (int v : todo) {
            const std::string res = base + "pm_result_" + std::to_string(v);
            for (int p = 0; p < np; ++p) remove((res + ".p" + std::to_string(p)).c_str());
            for (int g = 0; g < std::max(G, 64); ++g) {
                const std::string share = res + ".r" + std::to_string(g);
                if (remove(share.c_str()) != 0 && g >= G) continue;
                for (int p = 0; p < np; ++p) remove((share + ".p" + std::to_string(p)).c_str());
            }
        }
        fflush(stdout); fflush(stderr);
        std::vector<pid_t> pids;
        std::vector<int> pid_worker;
        for (int g = 0; g < G; ++g) {
            if (S.rank_off[(size_t)g + 1] == S.rank_off[(size_t)g]) continue;          // nothing for this worker
            const pid_t pid = fork();
            if (pid < 0) { fprintf(stderr, "[oc2pm] ERROR: fork failed\n"); failed = true; break; }
            if (pid == 0) {
                const PmTrace tr;
                necat_ctx* ctx = nullptr;
                if (necat_ctx_create(gpus[(size_t)g], &ctx)) { fprintf(stderr, "[oc2pm] ERROR: GPU %d: no usable gfx950 device (libnecat_hip has no CPU fallback)\n", gpus[(size_t)g]); _exit(1); }
                int status = 0;
                PmLanes lanes(gpus[(size_t)g]);            // a job's units on NECAT_PAIR_LANES contexts of this device (pm_job.h), kept from job to job
                // the reference volume of this worker's NEXT job is read from disk while the current job runs
                std::future<std::unique_ptr<PmLoaded>> ahead, cur;
                for (uint64_t k = S.rank_off[(size_t)g]; k < S.rank_off[(size_t)g + 1] && !status;) {
                    const int v = S.units[k].ref_vol;
                    std::vector<PmUnit> mine;
                    for (; k < S.rank_off[(size_t)g + 1] && S.units[k].ref_vol == v; ++k) mine.push_back(PmUnit{S.units[k].query_vol, S.units[k].slot_lo, S.units[k].slot_hi});
                    cur = std::move(ahead);
                    if (k < S.rank_off[(size_t)g + 1]) ahead = pm_load_async(vi, S.units[k].ref_vol);
                    char res[4096];
                    snprintf(res, sizeof res, "%spm_result_%d.r%d", base.c_str(), v, g);
                    fprintf(stdout, "Running %zu unit(s) of job 'oc2pmov %s %s %d %spm_result_%d' on GPU %d (worker %d)\n", mine.size(), options_to_string(&opt).c_str(), wrk_dir, v,
                            base.c_str(), v, gpus[(size_t)g], g);
                    fflush(stdout);
                    if ((status = pm_run_volume(ctx, vi, v, opt, res, "oc2pm", tr, cur.valid() ? &cur : nullptr, &mine, &lanes))) fprintf(stderr, "[oc2pm] ERROR: worker %d failed in the job of volume %d\n", g, v);
                }
                if (ahead.valid()) ahead.wait();
                lanes.close();
                necat_ctx_destroy(ctx);
                fflush(stdout); fflush(stderr);
                _exit(status ? 1 : 0);
            }
            pids.push_back(pid);
            pid_worker.push_back(g);
        }
        std::vector<uint8_t> worker_failed((size_t)G, 0);
        for (size_t w = 0; w < pids.size(); ++w) {
            int status = 0;
            if (waitpid(pids[w], &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) { failed = true; worker_failed[(size_t)pid_worker[w]] = 1; }
        }
        // the shares of a job, in worker order, make the job's file (written under a temporary name first, like the job itself does).  A volume whose
        // whole team succeeded is assembled and marked finished even when another worker failed (a rerun then only redoes the others, as the
        // whole-volume mode does); the shares of a volume with a failed worker are removed
        for (size_t t = 0; t < todo.size(); ++t) {
            const int v = todo[t];
            const std::string res = base + "pm_result_" + std::to_string(v), tmp = res + ".part";
            bool team_ok = true;
            for (int g = S.team_lo[(size_t)v]; g <= S.team_hi[(size_t)v]; ++g) {
                bool has = false;
                for (uint64_t k = S.rank_off[(size_t)g]; k < S.rank_off[(size_t)g + 1]; ++k) has = has || S.units[k].ref_vol == v;
                // (a worker stops at its first failing job: the jobs after it were not run either)
                if (has && worker_failed[(size_t)g]) team_ok = false;
            }
            if (!team_ok) {
                for (int g = S.team_lo[(size_t)v]; g <= S.team_hi[(size_t)v]; ++g) {
                    const std::string share = res + ".r" + std::to_string(g);
                    remove(share.c_str());
                    for (int p = 0; p < np; ++p) remove((share + ".p" + std::to_string(p)).c_str());
                }
                continue;
            }
            FILE* out = fopen(tmp.c_str(), "w");
            bool ok = out != nullptr;
            for (int g = S.team_lo[(size_t)v]; ok && g <= S.team_hi[(size_t)v]; ++g) {
                // (a worker of the team range that got no unit of v - its stretch rounded to nothing - wrote no file)
                bool has = false;
                for (uint64_t k = S.rank_off[(size_t)g]; k < S.rank_off[(size_t)g + 1]; ++k) has = has || S.units[k].ref_vol == v;
                if (has) ok = append_file(out, res + ".r" + std::to_string(g), false);
            }
            if (out && fclose(out) != 0) ok = false;
            for (int p = 0; ok && p < np; ++p) {
                FILE* po = nullptr;
                for (int g = S.team_lo[(size_t)v]; ok && g <= S.team_hi[(size_t)v]; ++g) {
                    bool has = false;
                    for (uint64_t k = S.rank_off[(size_t)g]; k < S.rank_off[(size_t)g + 1]; ++k) has = has || S.units[k].ref_vol == v;
                    if (!has) continue;
                    const std::string src = res + ".r" + std::to_string(g) + ".p" + std::to_string(p);
                    if (access(src.c_str(), F_OK) != 0) continue;
                    if (!po) { po = fopen((res + ".p" + std::to_string(p)).c_str(), "wb"); if (!po) { ok = false; break; } }
                    ok = append_file(po, src, false);
                    if (ok) remove(src.c_str());
                }
                if (po && fclose(po) != 0) ok = false;
            }
            if (ok) ok = rename(tmp.c_str(), res.c_str()) == 0;
            if (!ok) { fprintf(stderr, "[oc2pm] ERROR: assembling %s failed\n", res.c_str()); failed = true; continue; }
            for (int g = S.team_lo[(size_t)v]; g <= S.team_hi[(size_t)v]; ++g) remove((res + ".r" + std::to_string(g)).c_str());
            char fin[4096];
            snprintf(fin, sizeof fin, "%s/pm%d.finished", wrk_dir, v);
            FILE* f = fopen(fin, "w"); if (f) fclose(f);
        }
    } else if (!todo.empty()) {
        // the next volume to take, shared by the workers (forked before anything touches HIP)
        int* counter = (int*)mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        if (counter == MAP_FAILED) { fprintf(stderr, "[oc2pm] ERROR: mmap failed\n"); return 1; }
        *counter = 0;
        const size_t nworkers = std::min(gpus.size(), todo.size());
        // a worker that runs a single job is as short-lived as oc2pmov: same small band pools (see there)
        if (todo.size() <= gpus.size()) setenv("NECAT_BAND_POOL_MB", "1024", 0);
        fflush(stdout); fflush(stderr);
        std::vector<pid_t> pids;
        for (size_t g = 0; g < nworkers; ++g) {
            const pid_t pid = fork();
            if (pid < 0) { fprintf(stderr, "[oc2pm] ERROR: fork failed\n"); failed = true; break; }
            if (pid == 0) {
                const PmTrace tr;
                necat_ctx* ctx = nullptr;
                if (necat_ctx_create(gpus[g], &ctx)) { fprintf(stderr, "[oc2pm] ERROR: GPU %d: no usable gfx950 device (libnecat_hip has no CPU fallback)\n", gpus[g]); _exit(1); }
                tr.stage("context created");
                int status = 0;
                PmLanes lanes(gpus[g]);                    // a job's query volumes on NECAT_PAIR_LANES contexts of this device (pm_job.h), kept from job to job
                // While a job runs, the reference volume of the job this worker would draw NEXT (the first one nobody has claimed yet)
                // is read from disk; if another worker draws it first the read is thrown away.  Jobs are still claimed one at a time.
                std::future<std::unique_ptr<PmLoaded>> ahead, cur;
                int ahead_vol = -1;
                for (;;) {
                    const int idx = __atomic_fetch_add(counter, 1, __ATOMIC_RELAXED);
                    if (idx >= (int)todo.size()) break;
                    const int v = todo[idx];
                    cur = std::future<std::unique_ptr<PmLoaded>>();
                    if (ahead.valid()) { if (ahead_vol == v) cur = std::move(ahead); else { ahead.wait(); ahead = std::future<std::unique_ptr<PmLoaded>>(); } }
                    const int peek = __atomic_load_n(counter, __ATOMIC_RELAXED);
                    ahead_vol = peek < (int)todo.size() ? todo[peek] : -1;
                    if (ahead_vol >= 0) ahead = pm_load_async(vi, ahead_vol);
                    char res[4096], fin[4096];
                    snprintf(res, sizeof res, "%spm_result_%d", base.c_str(), v);
                    fprintf(stdout, "Running job 'oc2pmov %s %s %d %s' on GPU %d\n", options_to_string(&opt).c_str(), wrk_dir, v, res, gpus[g]);
                    fflush(stdout);
                    if ((status = pm_run_volume(ctx, vi, v, opt, res, "oc2pm", tr, cur.valid() ? &cur : nullptr, nullptr, &lanes))) { fprintf(stderr, "[oc2pm] ERROR: the job of volume %d failed\n", v); break; }
                    snprintf(fin, sizeof fin, "%s/pm%d.finished", wrk_dir, v);
                    FILE* f = fopen(fin, "w"); if (f) fclose(f);
                }
                if (ahead.valid()) ahead.wait();
                lanes.close();
                necat_ctx_destroy(ctx);
                fflush(stdout); fflush(stderr);
                _exit(status ? 1 : 0);
            }
            pids.push_back(pid);
        }
        for (pid_t pid : pids) {
            int status = 0;
            if (waitpid(pid, &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) failed = true;
        }
        munmap(counter, 4096);
    }
    if (failed) return 1;
    FILE* out = fopen(output, "w");
    if (!out) { fprintf(stderr, "[oc2pm] ERROR: cannot open %s\n", output); return 1; }
    bool wok = true;
    for (int i = 0; i < vi.num_volumes && wok; ++i) {
        char res[4096];
        snprintf(res, sizeof res, "%spm_result_%d", base.c_str(), i);
        FILE* in = fopen(res, "r");
        if (!in) { fprintf(stderr, "[oc2pm] ERROR: missing %s\n", res); fclose(out); return 1; }
        size_t k;
        while ((k = fread(buf.data(), 1, buf.size(), in)) > 0) if (fwrite(buf.data(), 1, k, out) != k) { wok = false; break; }
        if (ferror(in)) wok = false;
        fclose(in);
        if (wok) remove(res);
    }
    if (fclose(out) != 0) wok = false;
    // NECAT_PM_PARTITIONS=<batch size> (-j 0): the jobs also wrote their share of the consensus partitions (pm_job.h); merged
    // here into <output>.p<i> + <output>.partitions, the files oc2pcan would make from <output>
    const int pcan_batch = (opt.job == 0 && getenv("NECAT_PM_PARTITIONS")) ? atoi(getenv("NECAT_PM_PARTITIONS")) : 0;
    if (pcan_batch > 0 && wok) {
        const int np = (vi.num_reads + pcan_batch - 1) / pcan_batch;
        for (int p = 0; p < np && wok; ++p) {
            const std::string dst = std::string(output) + ".p" + std::to_string(p);
            FILE* po = fopen(dst.c_str(), "wb");
            if (!po) { wok = false; break; }
            for (int i = 0; i < vi.num_volumes && wok; ++i) {
                const std::string src = base + "pm_result_" + std::to_string(i) + ".p" + std::to_string(p);
                FILE* in = fopen(src.c_str(), "rb");
                if (!in) continue;                       // no record of that volume in this partition
                size_t k;
                while ((k = fread(buf.data(), 1, buf.size(), in)) > 0) if (fwrite(buf.data(), 1, k, po) != k) { wok = false; break; }
                if (ferror(in)) wok = false;
                fclose(in);
                if (wok) remove(src.c_str());
            }
            if (fclose(po) != 0) wok = false;
        }
        FILE* pn = fopen((std::string(output) + ".partitions").c_str(), "w");       // dump_num_partitions, pcan_aux.c:42-51
        if (!pn || fprintf(pn, "%d\n", np) < 0) wok = false;
        if (pn && fclose(pn) != 0) wok = false;
    }
    if (!wok) { fprintf(stderr, "[oc2pm] ERROR: writing %s failed\n", output); return 1; }
    return 0;
}
