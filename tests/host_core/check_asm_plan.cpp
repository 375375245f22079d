// check_asm_plan.cpp - GPU test of necat_asm_plan_batch (necat_amd/csrc/asm_plan.h: oc2asmpm's block vote and chained ranges on the device) against the
// host statement of the same steps (necat_amd/csrc/asm_core.h: Voter + BatchMapper::plan, which the CPU tests pin to the reference's own oc2asmpm), read by
// read: the planned (subject, strand) sequence (= the vote, the order and the cut) and every anchor (= the range).
//
//   check_asm_plan [map options] wrk_dir volume_id        exit 0 = identical; prints the first differences otherwise
//
// Links libnecat_hip.so (the table comes from the device's index build, downloaded in its sparse layout as the host vote reads it).
#include <stdio.h>
#include <stdlib.h>

#include <memory>

#include "../../necat_amd/csrc/asm_core.h"
#include "../../necat_amd/csrc/host_fmt.h"
#include "../../necat_amd/csrc/host_io.h"

using namespace necat;
using namespace necat_host;

struct HostCodes {
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

int main(int argc, char** argv)
{
    necat_map_options opt;
    necat_default_options(&opt);
    opt.num_candidates = opt.num_output = 100;
    if (argc < 3 || !parse_options(argc - 2, argv, &opt)) { fprintf(stderr, "usage: check_asm_plan [map options] wrk_dir volume_id\n"); return 2; }
    const char* wrk = argv[argc - 2];
    const int vid = atoi(argv[argc - 1]);
    std::string err;
    VolumesInfo vi;
    if (!load_volumes_info(wrk, &vi, &err)) { fprintf(stderr, "volumes: %s\n", err.c_str()); return 2; }
    HostVolume href;
    if (!load_volume(vi.names[vid].c_str(), &href, &err)) { fprintf(stderr, "volume: %s\n", err.c_str()); return 2; }
    necat_ctx* ctx = nullptr;
    if (necat_ctx_create(0, &ctx)) { fprintf(stderr, "no GPU\n"); return 2; }
    necat_volume* ref = nullptr; necat_index* ix = nullptr;
    if (necat_volume_upload(ctx, href.pac.data(), href.nbases, href.offset.data(), href.size.data(), href.offset.size(), &ref) ||
        necat_index_build(ctx, ref, opt.kmer_size, opt.kmer_cnt_cutoff, &ix)) { fprintf(stderr, "%s\n", necat_last_error(ctx)); return 2; }
    uint64_t n_table = 0, n_offsets = 0, n_pairs = 0, n_compact = 0;
    necat_index_size(ix, &n_table, &n_offsets);
    necat_index_sparse_size(ix, &n_pairs, &n_compact);
    std::unique_ptr<uint64_t[]> kmer_stats, pairs, compact, offset_list(new uint64_t[n_offsets + 1]);
    if (n_pairs) {
        pairs.reset(new uint64_t[2 * n_pairs]); compact.reset(new uint64_t[n_compact + 1]);
        if (necat_index_download_sparse(ctx, ix, pairs.get(), compact.get(), offset_list.get())) { fprintf(stderr, "%s\n", necat_last_error(ctx)); return 2; }
    } else {
        kmer_stats.reset(new uint64_t[n_table + 1]);
        if (necat_index_download(ctx, ix, kmer_stats.get(), offset_list.get())) { fprintf(stderr, "%s\n", necat_last_error(ctx)); return 2; }
    }
    HostCodes cref; cref.set(href);
    std::vector<uint64_t> ref_off(href.offset.size() + 1, 0);
    for (size_t i = 0; i < href.offset.size(); ++i) ref_off[i + 1] = href.offset[i] + href.size[i];
    asmpm::RefView rv;
    rv.seq_off = ref_off.data(); rv.nseq = href.offset.size();
    rv.kmer_list = [&](uint64_t h, uint64_t* n) -> const uint64_t* {
        uint64_t u;
        if (pairs) {
            const uint64_t bits = pairs[2 * (h >> 6)], bit = 1ULL << (h & 63);
            u = (bits & bit) ? compact[pairs[2 * (h >> 6) + 1] + (uint64_t)__builtin_popcountll(bits & (bit - 1))] : 0;
        } else u = kmer_stats[h];
        const uint64_t cnt = u >> 34, start = u & ((1ULL << 34) - 1);
        *n = cnt;
        return cnt ? offset_list.get() + start : nullptr;
    };
    auto subject_of = [&](int sid, int strand, std::vector<uint8_t>& s) { cref.strand((uint64_t)sid, strand, s); };
    const int ref_start = vi.read_start_id[vid];
    const char* dump = getenv("NECAT_ASM_DUMP_VOTES");
    uint64_t bad_reads = 0, total_plans = 0, total_anchors = 0, shown = 0;
    for (int v = vid; v < vi.num_volumes; ++v) {
        HostVolume own;
        const HostVolume* hreads = &href;
        necat_volume* reads = ref;
        HostCodes cown; const HostCodes* crd = &cref;
        if (v != vid) {
            if (!load_volume(vi.names[v].c_str(), &own, &err)) { fprintf(stderr, "volume: %s\n", err.c_str()); return 2; }
            hreads = &own; reads = nullptr;
            if (necat_volume_upload(ctx, own.pac.data(), own.nbases, own.offset.data(), own.size.data(), own.offset.size(), &reads)) { fprintf(stderr, "%s\n", necat_last_error(ctx)); return 2; }
            cown.set(own); crd = &cown;
        }
        const int read_start = vi.read_start_id[v];
        const uint64_t nreads = hreads->offset.size();
        if (dump) remove(dump);
        necat_asm_plan* plans = nullptr; uint64_t* first = nullptr;
        if (necat_asm_plan_batch(ctx, ix, ref, reads, read_start, ref_start, &opt, &plans, &first)) { fprintf(stderr, "necat_asm_plan_batch: %s\n", necat_last_error(ctx)); return 1; }
        // the device's ranked candidates, when the library dumped them
        std::vector<std::vector<int>> dev_votes(nreads);
        if (dump) if (FILE* f = fopen(dump, "rb")) {
            int hdr[3];
            while (fread(hdr, 4, 3, f) == 3) {
                std::vector<int>& d = dev_votes[(size_t)hdr[0]];
                d.resize((size_t)hdr[2] * 6 + 1);
                d[0] = hdr[1];
                if (hdr[2] && fread(d.data() + 1, 24, (size_t)hdr[2], f) != (size_t)hdr[2]) break;
            }
            fclose(f);
        }
        asmpm::Voter voter; voter.init(href.nbases);
        asmpm::BatchMapper mapper;
        std::vector<uint8_t> fwd, rev;
        std::vector<asmpm::VoteCandidate> votes;
        std::vector<asmpm::Planned> want;
        for (uint64_t r = 0; r < nreads; ++r) {
            crd->strand(r, 0, fwd); crd->strand(r, 1, rev);
            const int L = (int)fwd.size(), gid = (int)r + read_start;
            int64_t soff_max = INT32_MAX;
            if (gid >= ref_start && gid < ref_start + (int)rv.nseq) soff_max = (int64_t)ref_off[(size_t)(gid - ref_start)];
            votes.clear(); want.clear();
            voter.strand(fwd.data(), L, 0, (int)r, gid, ref_start, rv, opt.kmer_size, opt.scan_window, soff_max, votes);
            voter.strand(rev.data(), L, 1, (int)r, gid, ref_start, rv, opt.kmer_size, opt.scan_window, soff_max, votes);
            mapper.plan(votes, opt.num_candidates, fwd.data(), L, subject_of, want);          // (sorts `votes`)
            const uint64_t ng = first[r + 1] - first[r];
            const necat_asm_plan* got = plans + first[r];
            bool same_order = ng == want.size(), same = same_order;
            for (size_t k = 0; same_order && k < want.size(); ++k) same_order = got[k].sid - ref_start == want[k].sid && got[k].sdir == want[k].sdir && got[k].ssize == want[k].ssize;
            same = same_order;
            for (size_t k = 0; same && k < want.size(); ++k) same = got[k].qoff == want[k].qoff && (want[k].qoff < 0 || (got[k].soff == want[k].soff && got[k].score == want[k].score));
            total_plans += want.size();
            for (const asmpm::Planned& p : want) if (p.qoff >= 0) ++total_anchors;
            if (same) continue;
            ++bad_reads;
            if (shown >= 6) continue;
            ++shown;
            printf("read %lu of volume %d (length %d): %s differs; host votes %zu, host plan %zu, device plan %lu\n", (unsigned long)r, v, L, same_order ? "an ANCHOR" : "the PLAN ORDER (vote stage)",
                   votes.size(), want.size(), (unsigned long)ng);
            if (!same_order) {
                const std::vector<int>& d = dev_votes[(size_t)r];
                printf("  host votes (score chain target qstart tstart):"); for (size_t k = 0; k < votes.size() && k < 12; ++k) printf(" [%d %d %d %d %d]", votes[k].score, votes[k].chain, votes[k].target_id, votes[k].query_start, votes[k].target_start); printf("\n");
                if (!d.empty()) { printf("  dev  votes (%d in all)                      :", d[0]); for (size_t k = 0; k + 1 < d.size() / 6 + 1 && k < 12; ++k) printf(" [%d %d %d %d %d]", d[1 + 6 * k], d[2 + 6 * k], d[3 + 6 * k], d[4 + 6 * k], d[5 + 6 * k]); printf("\n"); }
            }
            for (size_t k = 0; k < std::max<size_t>(want.size(), ng) && k < 10; ++k) {
                printf("  [%zu] host:", k); if (k < want.size()) printf(" sid %d dir %d qoff %d soff %d score %d", want[k].sid, want[k].sdir, want[k].qoff, want[k].soff, want[k].score); else printf(" -");
                printf("   device:"); if (k < ng) printf(" sid %d dir %d qoff %d soff %d score %d", got[k].sid - ref_start, got[k].sdir, got[k].qoff, got[k].soff, got[k].score); else printf(" -");
                printf("\n");
            }
        }
        necat_free(plans); necat_free(first);
        if (reads != ref) necat_volume_free(ctx, reads);
    }
    printf("check_asm_plan: %lu planned pairs (%lu with an anchor), %lu reads differ\n", (unsigned long)total_plans, (unsigned long)total_anchors, (unsigned long)bad_reads);
    necat_index_free(ctx, ix); necat_volume_free(ctx, ref); necat_ctx_destroy(ctx);
    return bad_reads ? 1 : 0;
}
