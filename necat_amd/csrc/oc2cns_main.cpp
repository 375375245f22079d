// oc2cns - drop-in replacement of NECAT's oc2cns (consensus/main.c:51-78):
//   oc2cns [options] wrk_dir candidates cns_out raw_out [-mn node_id num_nodes]
// Same argv (consensus/cns_options.c:10, -mn as consensus/main.c:37-41), same inputs (the volume directory, the
// candidate partitions `candidates.p<i>` written by oc2pcan), same two FASTA outputs: corrected reads / stretches
// (cns_out) and the uncorrected rest (raw_out), record format DUMP_CNS_SEQ (common/cns_seq.h:24-44).
//
// Partitions go through two stages one partition apart: the extension loop of partition p + 1 on the GPU (a producer thread) beside the host
// consensus of partition p (consensus_one_partition.c:110 runs them one after the other; the files come out the same).
// Per partition: the extension loop of every template runs on the GPU (necat_cns_extension_batch: which candidates get
// aligned, which alignments count, with what weight - consensus/consensus_one_read.c:221-372), the consensus proper
// (tasc/) on the host threads (cns_consensus.h).  Records come out in template order (the reference's order with -t 1;
// with more threads the reference's order depends on scheduling).
// -r 1 (rescue_long_indels): candidates whose block-wise extension failed or fell short go through DALIGNER's local alignment and
// edlib's global path on the host threads after each device pass, as in the reference (cns_rescue.h).  -s 1 (small memory: reads
// loaded per partition) behaves like -s 0 - the read set is resident in HBM either way.  There is no CPU fallback for the
// block-wise alignments: without a usable GPU the program exits 1.
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "host_io.h"
#include "cns_consensus.h"

using namespace necat_host;

namespace {

struct CnsOpts {             // consensus/cns_options.h:6-18, defaults cns_options.c:10-22
    int min_align_size = 400, min_cov = 4, max_cov = 12, min_size = 500, full_consensus = 0;
    double error = 0.5, mapping_ratio = 0.8;
    int num_threads = 1, rescue_long_indels = 0, use_fixed_ident_cutoff = 0, small_memory = 0;
};

void describe(FILE* out)
{
    fprintf(out, "-a <Integer>\talign length cutoff\n-x <Integer>\tminimal coverage\n-y <Integer>\tmaximal coverage\n"
                 "-l <Integer>\tminimal length of corrected reads.\n-f <0 or 1>\tfull consensus or not: 1 = yes, 0 = no\n"
                 "-e <Real>\tsequencing error\n-p <Real>\tminimal mapping ratio\n-t <Integer>\tnumber of cpu threads\n"
                 "-r <0 or 1>\trescue long indels or not: 1 = yes, 0 = no\n-u <0 or 1>\tuse dynamic or fixed ident cutoff: 1 = fixed, 0 = dynamic\n"
                 "-s <0 or 1>\tuse small memoty\nDEFAULT OPTIONS:\n");
    CnsOpts d;
    fprintf(out, "-a %d -x %d -y %d -l %d -f %d -e %f -p %f -t %d -r %d -u %d -s %d\t\n", d.min_align_size, d.min_cov, d.max_cov, d.min_size, d.full_consensus,
            d.error, d.mapping_ratio, d.num_threads, d.rescue_long_indels, d.use_fixed_ident_cutoff, d.small_memory);
}

bool parse(int argc, char** argv, CnsOpts* o)
{
    optind = 1;
    int c;
    while ((c = getopt(argc, argv, "a:x:y:l:f:e:p:t:r:u:s:")) != -1) {
        switch (c) {
        case 'a': o->min_align_size = atoi(optarg); break;
        case 'x': o->min_cov = atoi(optarg); break;
        case 'y': o->max_cov = atoi(optarg); break;
        case 'l': o->min_size = atoi(optarg); break;
        case 'f': o->full_consensus = atoi(optarg); break;
        case 'e': o->error = atof(optarg); break;
        case 'p': o->mapping_ratio = atof(optarg); break;
        case 't': o->num_threads = atoi(optarg); break;
        case 'r': o->rescue_long_indels = atoi(optarg); break;
        case 'u': o->use_fixed_ident_cutoff = atoi(optarg); break;
        case 's': o->small_memory = atoi(optarg); break;
        default: return false;
        }
    }
    return true;
}

int usage(const char* prog)
{
    fprintf(stderr, "USAGE:\n%s [options] wrk_dir candidates cns_out raw_out\n\nIf Multiple Nodes Are Used:\n"
                    "%s [options] wrk_dir candidates cns_out raw_out -mn node_id num_nodes\n\nOPTIONS AND DESCRIPTIONS:\n", prog, prog);
    describe(stderr);
    return 1;
}

int fail(const char* what, const char* detail) { fprintf(stderr, "[oc2cns] ERROR: %s: %s\n", what, detail); return 1; }

// all volumes as one read set with global ids (merge_volumes, common/makedb_aux.c:137-153)
struct ReadSet {
    std::vector<uint8_t> codes;            // one byte per base (0..3)
    std::vector<uint64_t> offset, size;
    std::vector<std::string> names;
    const uint8_t* read(uint64_t id) const { return codes.data() + offset[id]; }
};

}  // namespace

int main(int argc, char** argv)
{
    necat_host::necat_cli_env();          // (before the first HIP call: host_io.h)
    CnsOpts opt;
    int spid = 0, nnode = 1;
    int ac = argc;
    if (ac >= 8 && strcmp(argv[ac - 3], "-mn") == 0) { spid = atoi(argv[ac - 2]); nnode = atoi(argv[ac - 1]); ac -= 3; }
    if (ac < 5 || !parse(ac - 4, argv, &opt) || nnode < 1 || spid < 0) return usage(argv[0]);
    const char* wrk_dir = argv[ac - 4];
    const std::string can_path = argv[ac - 3];
    const char* cns_out_path = argv[ac - 2];
    const char* raw_out_path = argv[ac - 1];
    if (opt.max_cov < 1 || opt.min_cov < 0) return fail("options", "coverage limits out of range");

    std::string err;
    VolumesInfo vi;
    if (!load_volumes_info(wrk_dir, &vi, &err)) return fail("volume directory", err.c_str());
    ReadSet rs;
    {
        log_line("", "load reads");
        const double t0 = now_sec();
        uint64_t total = 0;
        std::vector<HostVolume> vols((size_t)vi.num_volumes);
        for (int v = 0; v < vi.num_volumes; ++v) {
            if (!load_volume(vi.names[v].c_str(), &vols[v], &err)) return fail("volume", err.c_str());
            total += vols[v].nbases;
        }
        rs.codes.resize(total);
        uint64_t at = 0;
        for (int v = 0; v < vi.num_volumes; ++v) {
            const HostVolume& hv = vols[v];
            for (uint64_t i = 0; i < hv.offset.size(); ++i) {
                rs.offset.push_back(at + hv.offset[i]); rs.size.push_back(hv.size[i]); rs.names.emplace_back(hv.name(i));
            }
            uint8_t* dst = rs.codes.data() + at;
            for (uint64_t i = 0; i < hv.nbases; ++i) dst[i] = (uint8_t)((hv.pac[i >> 2] >> ((~i & 3) << 1)) & 3);
            at += hv.nbases;
        }
        log_line("[%s] INFO: '%s' takes %.2lf secs.\n", "load reads", now_sec() - t0);
    }
    const uint64_t nreads = rs.size.size();

    int num_partitions = 0;
    {
        FILE* in = fopen((can_path + ".partitions").c_str(), "r");
        if (!in || fscanf(in, "%d", &num_partitions) != 1) return fail("candidates", "cannot read the .partitions file (run oc2pcan first)");
        fclose(in);
    }
    FILE* cns_out = fopen(cns_out_path, "w");
    FILE* raw_out = fopen(raw_out_path, "w");
    if (!cns_out || !raw_out) return fail("output", "cannot open for writing");

    necat_ctx* ctx = nullptr;
    const char* dev_env = getenv("NECAT_GPU");
    if (necat_ctx_create(dev_env ? atoi(dev_env) : 0, &ctx)) return fail("GPU", "no usable gfx950 device (libnecat_hip has no CPU fallback)");
    necat_volume* reads = nullptr;
    {
        // NECAT pac of the merged set (first base of a byte in its top two bits)
        std::vector<uint8_t> pac((rs.codes.size() + 3) / 4 + 8, 0);
        for (uint64_t i = 0; i < rs.codes.size(); ++i) pac[i >> 2] |= (uint8_t)(rs.codes[i] << ((~i & 3) << 1));
        if (necat_volume_upload(ctx, pac.data(), rs.codes.size(), rs.offset.data(), rs.size.data(), nreads, &reads)) return fail("necat_volume_upload", necat_last_error(ctx));
    }
    necat_cns_options co; necat_cns_default_options(&co);
    co.min_align_size = opt.min_align_size; co.min_cov = opt.min_cov; co.max_cov = opt.max_cov; co.error = opt.error;
    co.mapping_ratio = opt.mapping_ratio; co.use_fixed_ident_cutoff = opt.use_fixed_ident_cutoff;
    co.rescue_long_indels = opt.rescue_long_indels != 0;
    const int nthreads = std::max(1, std::min(opt.num_threads, 256));       // -t: host threads of the consensus proper

    // ---- two stages, one partition apart (round 6): a producer thread reads partition p + 1 and runs its extension loop on the device
    // (necat_cns_load_partition + necat_cns_extension_batch: the context is that thread's alone from here on) while the host threads do the
    // consensus proper of partition p - 88 % of a partition's time is host work, the device loop of the next one hides behind it.
    // One finished partition waits at most (the alignment columns of a partition are hundreds of MB of pinned memory).  Output order =
    // partition order, as before.  NECAT_CNS_PIPELINE=0: one partition after the other on the main thread (the round-5 form).
    struct PartWork {
        int pid = -1; bool empty = false; int rc = 0; std::string what, detail;
        necat_candidate* cands = nullptr; uint64_t* tmpl_off = nullptr; uint64_t* n_all = nullptr; uint64_t nt = 0;
        necat_cns_result* res = nullptr; double t_gpu = 0, t0 = 0;
    };
    auto device_stage = [&](int pid) -> PartWork {
        PartWork w; w.pid = pid; w.t0 = now_sec();
        std::vector<uint8_t> packed;
        {
            char suffix[32];
            snprintf(suffix, sizeof suffix, ".p%d", pid);
            FILE* in = fopen((can_path + suffix).c_str(), "rb");
            if (!in) { w.rc = 1; w.what = "candidates"; w.detail = "cannot open " + can_path + suffix; return w; }
            fseek(in, 0, SEEK_END);
            const long bytes = ftell(in);
            fseek(in, 0, SEEK_SET);
            packed.resize((size_t)(bytes / 28) * 28);
            if (!packed.empty() && fread(packed.data(), 1, packed.size(), in) != packed.size()) { fclose(in); w.rc = 1; w.what = "candidates"; w.detail = "short read"; return w; }
            fclose(in);
        }
        if (packed.empty()) { w.empty = true; return w; }
        if (necat_cns_load_partition(ctx, reads, packed.data(), packed.size() / 28, &w.cands, &w.tmpl_off, &w.n_all, &w.nt)) { w.rc = 1; w.what = "necat_cns_load_partition"; w.detail = necat_last_error(ctx); return w; }
        if (necat_cns_extension_batch(ctx, reads, w.cands, w.tmpl_off, w.n_all, w.nt, &co, &w.res)) { w.rc = 1; w.what = "necat_cns_extension_batch"; w.detail = necat_last_error(ctx); return w; }
        w.t_gpu = now_sec() - w.t0;
        return w;
    };
    std::vector<int> pids;
    for (int pid = spid; pid < num_partitions; pid += nnode) pids.push_back(pid);
    const bool pipelined = pids.size() > 1 && !(getenv("NECAT_CNS_PIPELINE") && atoi(getenv("NECAT_CNS_PIPELINE")) == 0);
    std::mutex qmu; std::condition_variable qcv;
    std::deque<PartWork> ready;              // finished device stages, in partition order (at most one waits)
    bool stop = false;
    std::thread producer;
    if (pipelined) producer = std::thread([&]() {
        for (int pid : pids) {
            { std::unique_lock<std::mutex> lk(qmu); qcv.wait(lk, [&] { return ready.size() < 1 || stop; }); if (stop) return; }
            PartWork w = device_stage(pid);
            const bool failed = w.rc != 0;
            { std::lock_guard<std::mutex> lk(qmu); ready.push_back(std::move(w)); }
            qcv.notify_all();
            if (failed) return;
        }
    });
    auto stop_producer = [&]() { if (producer.joinable()) { { std::lock_guard<std::mutex> lk(qmu); stop = true; } qcv.notify_all(); producer.join(); } };
    double t_prev_done = now_sec();
    for (size_t pi = 0; pi < pids.size(); ++pi) {
        const int pid = pids[pi];
        char job[128];
        snprintf(job, sizeof job, "consensus partition %d", pid);
        log_line("", job);
        PartWork W;
        if (pipelined) {
            std::unique_lock<std::mutex> lk(qmu);
            qcv.wait(lk, [&] { return !ready.empty(); });
            W = std::move(ready.front()); ready.pop_front();
            lk.unlock(); qcv.notify_all();
        } else W = device_stage(pid);
        if (W.rc) { stop_producer(); return fail(W.what.c_str(), W.detail.c_str()); }
        const double t0 = pipelined ? t_prev_done : W.t0;          // what this partition added to the wall clock
        if (W.empty) { log_line("[%s] INFO: '%s' takes %.2lf secs.\n", job, now_sec() - t0); t_prev_done = now_sec(); continue; }
        necat_candidate* cands = W.cands; uint64_t* tmpl_off = W.tmpl_off; uint64_t* n_all = W.n_all; const uint64_t nt = W.nt;
        necat_cns_result* res = W.res;
        const double t_gpu = W.t_gpu;
        const double t_host0 = now_sec();
        // ---- consensus proper, templates in parallel on the host
        std::vector<std::string> out_cns((size_t)nt), out_raw((size_t)nt);
        std::vector<uint8_t> corrected((size_t)nt, 0);
        std::atomic<uint64_t> next(0);
        auto worker = [&]() {
            cns::Worker w;
            std::vector<cns::OverlapIn> ovs;
            for (;;) {
                const uint64_t t = next.fetch_add(1);
                if (t >= nt) break;
                const necat_cns_template& T = res->templates[t];
                if (!T.examined) continue;
                const necat_candidate& c0 = cands[tmpl_off[t]];
                const int tid = c0.sid, tsize = (int)c0.ssize;
                ovs.clear();
                for (uint64_t k = T.ovlp_begin; k < T.ovlp_end; ++k) {
                    const necat_cns_overlap& ov = res->overlaps[k];
                    const necat_candidate& c = cands[ov.cand];
                    cns::OverlapIn o;
                    o.ops = res->ops[ov.ops_block] + ov.ops_off; o.ncols = ov.align_size; o.toff = ov.toff; o.weight = ov.weight;
                    o.qfwd = rs.read((uint64_t)c.qid); o.qsize = (int)c.qsize; o.qoff = ov.qoff; o.qdir = c.qdir;
                    ovs.push_back(o);
                }
                corrected[t] = cns::consensus_template(w, ovs.data(), ovs.size(), rs.read((uint64_t)tid), tsize, tid, rs.names[(size_t)tid].c_str(), opt.min_cov,
                                                       opt.min_size, opt.full_consensus != 0, T.num_can, T.num_ovlps, T.ident_cutoff, out_cns[t], out_raw[t]) ? 1 : 0;
            }
        };
        {
            std::vector<std::thread> pool;
            for (int i = 1; i < nthreads; ++i) pool.emplace_back(worker);
            worker();
            for (auto& th : pool) th.join();
        }
        bool wok = true;
        for (uint64_t t = 0; t < nt; ++t) {
            if (!out_cns[t].empty()) wok = wok && fwrite(out_cns[t].data(), 1, out_cns[t].size(), cns_out) == out_cns[t].size();
            if (!out_raw[t].empty()) wok = wok && fwrite(out_raw[t].data(), 1, out_raw[t].size(), raw_out) == out_raw[t].size();
        }
        // the reads of the partition's id range nobody corrected go out whole (consensus_one_partition.c:172-194; the last id
        // of the range is left out there: `i < max_read_id`) - only with -s 0: the reference does this inside `if (reads)`, and with
        // -s 1 it never loads `reads` (the small-memory path reads sequences per partition), so it writes none of them
        if (!opt.small_memory) {
            std::vector<uint8_t> done;
            int min_id = cands[tmpl_off[0]].sid, max_id = min_id;
            for (uint64_t t = 0; t < nt; ++t) { const int id = cands[tmpl_off[t]].sid; min_id = std::min(min_id, id); max_id = std::max(max_id, id); }
            done.assign((size_t)(max_id - min_id + 1), 0);
            for (uint64_t t = 0; t < nt; ++t) if (corrected[t]) done[(size_t)(cands[tmpl_off[t]].sid - min_id)] = 1;
            std::string rec;
            for (int id = min_id; id < max_id; ++id) {
                if (done[(size_t)(id - min_id)]) continue;
                rec.clear();
                cns::uncorrected_record(rec, rs.read((uint64_t)id), (int)rs.size[(size_t)id], id, rs.names[(size_t)id].c_str());
                wok = wok && fwrite(rec.data(), 1, rec.size(), raw_out) == rec.size();
            }
        }
        if (!wok) { stop_producer(); return fail("output", "write failed"); }
        necat_cns_result_free(res);
        necat_free(cands); necat_free(tmpl_off); necat_free(n_all);
        fprintf(stdout, "[oc2cns] partition %d: %lu templates, extension loop %.2f s%s, consensus %.2f s (%d host threads)\n", pid, (unsigned long)nt, t_gpu,
                pipelined && pi ? " (beside the previous partition's consensus)" : "", now_sec() - t_host0, nthreads);
        log_line("[%s] INFO: '%s' takes %.2lf secs.\n", job, now_sec() - t0);
        t_prev_done = now_sec();
    }
    stop_producer();
    const bool ok = fclose(cns_out) == 0;
    const bool ok2 = fclose(raw_out) == 0;
    if (!ok || !ok2) return fail("output", "write failed");
    necat_volume_free(ctx, reads);
    necat_ctx_destroy(ctx);
    return 0;
}
