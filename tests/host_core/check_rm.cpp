// check_rm.cpp - CPU test of the host side of necat_map_reference (necat_amd/csrc/rm_host.h + rm_window in ext_core.h): the
// candidates come from the ORACLE's seeding, every candidate is aligned against its stretch of the reference with the ORACLE's
// onc_align (the stand-in for the device pass, records built as k_ext_result builds them), and rm_host.h replays the read's
// candidates - containment, drop, rescue pair.  The records are written as the reference's oc2rm_worker -i 0 writes them, for
// the test to compare with the output of the REFERENCE's own program (oracle/_ref/oc2rm_worker -t 1).
//
//   check_rm wrk_dir reference out [map options as oc2rm_worker takes them]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../necat_amd/csrc/rm_host.h"
#include "../../oracle/necat_oracle.h"

using namespace necat;

static std::vector<uint64_t> words_of(const ora_volume& v, std::vector<uint64_t>& off)
{
    std::vector<uint64_t> w((v.nbases + 31) / 32 + 1, 0);
    off.assign(v.nseq + 1, 0);
    std::vector<uint8_t> buf;
    for (uint64_t i = 0; i < v.nseq; ++i) {
        off[i + 1] = off[i] + v.size[i];
        buf.resize(v.size[i] + 1);
        ora_volume_extract(&v, i, 0, buf.data());
        for (uint64_t k = 0; k < v.size[i]; ++k) { const uint64_t g = off[i] + k; w[g >> 5] |= (uint64_t)buf[k] << ((g & 31) * 2); }
    }
    return w;
}

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    ora_options opt;
    ora_options_default(&opt);
    opt.kmer_size = 15; opt.scan_window = 5; opt.kmer_cnt_cutoff = 500; opt.block_size = 1000; opt.block_score_cutoff = 3;      // map_options.c:31-46
    opt.num_candidates = 20; opt.align_size_cutoff = 400; opt.ddfs_cutoff = 0.25; opt.error = 0.5; opt.num_output = 20; opt.job = 1;
    if (ora_options_parse(argc - 3, argv, &opt)) return 2;
    const char* wrk = argv[argc - 3];
    ora_volume ref;
    if (ora_volume_load(argv[argc - 2], &ref)) return 2;
    FILE* out = fopen(argv[argc - 1], "w");
    if (!out) return 2;
    ora_volumes_info vi;
    if (ora_volumes_info_load(wrk, &vi)) return 2;
    ora_index* ix = ora_index_build(&ref, opt.kmer_size, opt.kmer_cnt_cutoff);
    std::vector<uint64_t> ref_off, ref_w = words_of(ref, ref_off);
    ora_wfd* w = ora_wfd_new(ref.nbases, opt.block_size, opt.kmer_size, opt.block_score_cutoff);
    ora_aligner* al = ora_aligner_new(opt.error);
    const rescue::DalignSpec spec = rescue::spec_for_error(opt.error);
    rm::Worker wk(spec, opt.error);
    uint64_t n_records = 0, n_cands = 0;
    for (int v = 0; v < vi.num_volumes; ++v) {
        ora_volume reads;
        if (ora_volume_load(vi.names[v], &reads)) return 2;
        std::vector<uint64_t> rd_off, rd_w = words_of(reads, rd_off);
        rm::Words hr, hf;
        hr.w = rd_w.data(); hr.seq_off = rd_off.data(); hf.w = ref_w.data(); hf.seq_off = ref_off.data();
        const int read_start = vi.read_start_id[v];
        ora_can_vec cans = {0, 0, 0};
        std::vector<uint8_t> fwd, rev, sub;
        std::vector<necat_candidate> cs;
        std::vector<necat_m4> m4, acc;
        std::vector<uint8_t> ok;
        for (uint64_t i = 0; i < reads.nseq; ++i) {
            const size_t L = reads.size[i];
            fwd.resize(L + 1); rev.resize(L + 1);
            cans.n = 0;
            ora_volume_extract(&reads, i, 0, fwd.data());
            ora_find_candidates(fwd.data(), (int)L, (int)i, 0, read_start, 0, 0, &ref, ix, &opt, w, &cans);
            ora_volume_extract(&reads, i, 1, rev.data());
            ora_find_candidates(rev.data(), (int)L, (int)i, 1, read_start, 0, 0, &ref, ix, &opt, w, &cans);
            std::sort(cans.a, cans.a + cans.n, [](const ora_candidate& a, const ora_candidate& b) {      // GappedCandidate_RmScoreGT, rm_worker.c:15-24
                if (a.score != b.score) return a.score > b.score;
                if (a.qdir != b.qdir) return a.qdir < b.qdir;
                if (a.sid != b.sid) return a.sid < b.sid;
                if (a.qoff != b.qoff) return a.qoff < b.qoff;
                return a.soff < b.soff;
            });
            if (cans.n > (size_t)opt.num_candidates) cans.n = (size_t)opt.num_candidates;
            cs.resize(cans.n); m4.resize(cans.n); ok.resize(cans.n);
            for (size_t k = 0; k < cans.n; ++k) {
                const ora_candidate& o = cans.a[k];
                necat_candidate c;
                memset(&c, 0, sizeof c);
                c.qid = o.qid + read_start; c.sid = o.sid; c.qdir = o.qdir; c.sdir = o.sdir; c.score = o.score;
                c.qbeg = (uint64_t)o.qbeg; c.qend = (uint64_t)o.qend; c.qsize = (uint64_t)o.qsize; c.sbeg = (uint64_t)o.sbeg; c.send = (uint64_t)o.send;
                c.ssize = (uint64_t)o.ssize; c.qoff = (uint64_t)o.qoff; c.soff = (uint64_t)o.soff;
                cs[k] = c;
                // the device pass: the block-wise alignment against the stretch, the record of k_ext_result
                int64_t from, to, woff;
                rm_window(o.qoff, o.qsize, o.soff, o.ssize, &from, &to, &woff);
                hf.decode(o.sid, 0, from, to, sub);
                ora_align_result r;
                ok[k] = (uint8_t)ora_onc_align(al, o.qdir ? rev.data() : fwd.data(), (int)o.qoff, (int)o.qsize, sub.data(), (int)woff, (int)(to - from), 512,
                                               opt.align_size_cutoff, 1, &r);
                necat_m4 m;
                memset(&m, 0, sizeof m);
                m.qid = c.qid; m.qdir = c.qdir; m.qoff = (uint64_t)r.qoff; m.qend = (uint64_t)r.qend; m.qext = c.qoff; m.qsize = c.qsize;
                m.sid = c.sid; m.sdir = 0; m.soff = (uint64_t)(r.toff + from); m.send = (uint64_t)(r.tend + from); m.sext = c.soff; m.ssize = c.ssize;
                m.ident_perc = r.ident_perc; m.vscore = c.score;
                if (m.qdir == 1) { const uint64_t qo = m.qsize - m.qend, qe = m.qsize - m.qoff; m.qoff = qo; m.qend = qe; }
                m4[k] = m;
            }
            n_cands += cans.n;
            acc.clear();
            wk.replay(cs.data(), m4.data(), ok.data(), 0, cs.size(), hr, hf, read_start, 0, opt.align_size_cutoff, acc);
            for (const necat_m4& m : acc) {
                fprintf(out, "%d\t%d\t%.2f\t%d\t%d\t%lu\t%lu\t%lu\t%d\t%lu\t%lu\t%lu\n", m.qid, m.sid, m.ident_perc, m.vscore, m.qdir, (unsigned long)m.qoff,
                        (unsigned long)m.qend, (unsigned long)m.qsize, m.sdir, (unsigned long)m.soff, (unsigned long)m.send, (unsigned long)m.ssize);
                ++n_records;
            }
        }
        free(cans.a);
        ora_volume_free(&reads);
    }
    fclose(out);
    printf("records=%lu candidates=%lu rescue_tried=%lu rescued=%lu\n", (unsigned long)n_records, (unsigned long)n_cands, (unsigned long)wk.n_rescue_tried,
           (unsigned long)wk.n_rescued);
    return 0;
}
