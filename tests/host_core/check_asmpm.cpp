// check_asmpm.cpp - CPU test of necat_amd/csrc/asm_core.h (oc2asmpm restated: block vote, MEM chain, 2048-bp block extension with
// DALIGNER end extension) behind the ORACLE's volume reader, lookup table and block aligner (ora_onc_align with 2048-bp blocks and
// tail match length 8 = the clone in asm_pm/blockwise_edlib.c).  Writes what the reference's oc2asmpm -u 0 writes, for the test to
// compare with the output of the REFERENCE's own program (oracle/_ref/oc2asmpm -t 1).
//
//   check_asmpm [map options] wrk_dir volume_id out
//
// CHECK_ASM_BATCH=1 in the environment: the two-phase walk the program uses (BatchMapper: plan all anchors, align them, finish) instead of the
// candidate-by-candidate one.
#include <stdio.h>
#include <stdlib.h>

#include <memory>

#include "../../necat_amd/csrc/asm_core.h"
#include "../../oracle/necat_oracle.h"

using namespace necat;

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    ora_options opt;
    ora_options_default(&opt);
    opt.num_candidates = opt.num_output = 100;          // asmpm.c:14 (MAXC)
    if (ora_options_parse(argc - 3, argv, &opt)) return 2;
    const char* wrk = argv[argc - 3];
    const int vid = atoi(argv[argc - 2]);
    FILE* out = fopen(argv[argc - 1], "w");
    if (!out) return 2;
    ora_volumes_info vi;
    if (ora_volumes_info_load(wrk, &vi)) return 2;
    ora_volume ref;
    if (ora_volume_load(vi.names[vid], &ref)) return 2;
    const int ref_start = vi.read_start_id[vid];
    ora_index* ix = ora_index_build(&ref, opt.kmer_size, opt.kmer_cnt_cutoff);
    std::vector<uint64_t> ref_off(ref.nseq + 1, 0);
    for (uint64_t i = 0; i < ref.nseq; ++i) ref_off[i + 1] = ref_off[i] + ref.size[i];
    asmpm::RefView rv;
    rv.seq_off = ref_off.data(); rv.nseq = ref.nseq;
    rv.kmer_list = [&](uint64_t h, uint64_t* n) -> const uint64_t* {
        const uint64_t u = ix->kmer_stats[h], cnt = u >> 34, start = u & ((1ULL << 34) - 1);
        *n = cnt;
        return cnt ? ix->offset_list + start : nullptr;
    };
    ora_aligner* al = ora_aligner_new(0.5);
    asmpm::BlockAlignFn block_align = [&](const uint8_t* read, int qoff, int qsize, const uint8_t* subject, int soff, int ssize, int min_align, asmpm::BlockAlignment* a) {
        ora_align_result r;
        if (!ora_onc_align(al, read, qoff, qsize, subject, soff, ssize, 2048, min_align, 8, &r)) return false;
        a->qoff = r.qoff; a->qend = r.qend; a->toff = r.toff; a->tend = r.tend; a->ident_perc = r.ident_perc;
        a->qaln.assign(r.query_align, (size_t)r.align_size); a->taln.assign(r.target_align, (size_t)r.align_size);
        return true;
    };
    auto subject_of = [&](int sid, int strand, std::vector<uint8_t>& s) { s.resize(ref.size[sid] + 1); ora_volume_extract(&ref, (uint64_t)sid, strand, s.data()); s.resize(ref.size[sid]); };
    asmpm::Voter voter;
    voter.init(ref.nbases);
    asmpm::ReadMapper mapper;
    asmpm::BatchMapper batch;
    const int batch_mode = getenv("CHECK_ASM_BATCH") ? atoi(getenv("CHECK_ASM_BATCH")) : 0;       // 2: the finish step on packed columns (Extender::ends_packed), as the program runs it
    const bool use_batch = batch_mode != 0;
    uint64_t n_records = 0, n_votes = 0;
    for (int v = vid; v < vi.num_volumes; ++v) {
        ora_volume reads;
        if (ora_volume_load(vi.names[v], &reads)) return 2;
        const int read_start = vi.read_start_id[v];
        std::vector<uint8_t> fwd, rev;
        std::vector<asmpm::VoteCandidate> votes;
        std::vector<necat_m4> recs;
        for (uint64_t i = 0; i < reads.nseq; ++i) {
            const int L = (int)reads.size[i], gid = (int)i + read_start;
            fwd.resize((size_t)L + 1); rev.resize((size_t)L + 1);
            ora_volume_extract(&reads, i, 0, fwd.data());
            ora_volume_extract(&reads, i, 1, rev.data());
            int64_t soff_max = INT32_MAX;
            if (gid >= ref_start && gid < ref_start + (int)ref.nseq) soff_max = (int64_t)ref_off[(size_t)(gid - ref_start)];
            votes.clear();
            voter.strand(fwd.data(), L, 0, (int)i, gid, ref_start, rv, opt.kmer_size, opt.scan_window, soff_max, votes);
            voter.strand(rev.data(), L, 1, (int)i, gid, ref_start, rv, opt.kmer_size, opt.scan_window, soff_max, votes);
            n_votes += votes.size();
            recs.clear();
            if (!use_batch) mapper.go(votes, opt.num_candidates, fwd.data(), (int)i, L, subject_of, block_align, recs);
            else {
                std::vector<asmpm::Planned> pl;
                batch.plan(votes, opt.num_candidates, fwd.data(), L, subject_of, pl);
                std::vector<asmpm::BlockAlignment> ba(pl.size());
                std::unique_ptr<bool[]> ok(new bool[pl.size() + 1]);
                std::vector<uint8_t> subj;
                for (size_t k = 0; k < pl.size(); ++k) {
                    ok[k] = false;
                    if (pl[k].qoff < 0) continue;
                    subject_of(pl[k].sid, pl[k].sdir, subj);
                    ok[k] = block_align(fwd.data(), pl[k].qoff, L, subj.data(), pl[k].soff, pl[k].ssize, 400, &ba[k]);
                }
                if (batch_mode == 2) {
                    // the columns two bits each, the strings dropped: what necat_asm_align_batch hands the program
                    std::vector<std::vector<uint8_t>> packed(pl.size());
                    std::vector<const uint8_t*> ops(pl.size(), nullptr);
                    std::vector<size_t> ncols(pl.size(), 0);
                    for (size_t k = 0; k < pl.size(); ++k) {
                        if (!ok[k]) continue;
                        const size_t n = ba[k].qaln.size();
                        packed[k].assign((n + 3) / 4 + 1, 0);
                        for (size_t c = 0; c < n; ++c) { const int op = ba[k].qaln[c] == '-' ? 2 : ba[k].taln[c] == '-' ? 1 : 0; packed[k][c >> 2] |= (uint8_t)(op << ((c & 3) * 2)); }
                        ops[k] = packed[k].data(); ncols[k] = n;
                        ba[k].qaln.clear(); ba[k].taln.clear();
                    }
                    batch.finish_packed(pl.data(), pl.size(), ok.get(), ba.data(), ops.data(), ncols.data(), fwd.data(), (int)i, L, subject_of, recs);
                } else
                batch.finish(pl.data(), pl.size(), ok.get(), ba.data(), fwd.data(), (int)i, L, subject_of, recs);
            }
            for (const necat_m4& m : recs) {        // DUMP_ASM_M4_HDR_ID (m4_record.h:99-124)
                fprintf(out, "%s\t%s\t%.2f\t%d\t%d\t%lu\t%lu\t%lu\t%d\t%lu\t%lu\t%lu\n", reads.hdr + reads.hdr_offset[m.qid], ref.hdr + ref.hdr_offset[m.sid], m.ident_perc,
                        m.vscore, m.qdir, (unsigned long)m.qoff, (unsigned long)m.qend, (unsigned long)m.qsize, m.sdir, (unsigned long)m.soff, (unsigned long)m.send,
                        (unsigned long)m.ssize);
                ++n_records;
            }
        }
        ora_volume_free(&reads);
    }
    fclose(out);
    printf("records=%lu votes=%lu\n", (unsigned long)n_records, (unsigned long)n_votes);
    return 0;
}
