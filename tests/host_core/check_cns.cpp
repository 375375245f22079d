// check_cns.cpp - CPU test of necat_amd/csrc/cns_loop.h (the host side of necat_cns_extension_batch): the
// select / replay logic is run with the ORACLE's onc_align plugged in as the aligner and compared, template
// by template, with the oracle's sequential restatement of the reference loop (oracle/cns_oracle.c).
// The oracle appears here only as the checker and as a stand-in aligner for a machine without a GPU.
//
//   check_cns wrk_dir can_prefix [min_align min_cov max_cov error ratio fixed [spec_extra spec_cover [adapt_mult adapt_min [rescue log]]]]
//
// rescue = 1: cns_rescue.h runs behind the stand-in aligner as it does behind the device pass in the library (oc2cns -r 1).  The
// oracle's loop has no rescue, so nothing is compared in-process; `log` receives the loop's decisions in the format of
// oracle/cns_ref_harness.c ("full"), for the test to compare with the REFERENCE's own log.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../necat_amd/csrc/cns_loop.h"
#include "../../necat_amd/csrc/cns_rescue.h"
#include "../../oracle/necat_oracle.h"

using namespace necat;

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage\n"); return 2; }
    necat_cns_options opt = {400, 4, 12, 0.5, 0.8, 0, 0};
    cns::Knobs kn;
    if (argc >= 9) {
        opt.min_align_size = atoi(argv[3]); opt.min_cov = atoi(argv[4]); opt.max_cov = atoi(argv[5]);
        opt.error = atof(argv[6]); opt.mapping_ratio = atof(argv[7]); opt.use_fixed_ident_cutoff = atoi(argv[8]);
    }
    if (argc >= 11) { kn.spec_estimate_extra = atoi(argv[9]); kn.spec_cover = atoi(argv[10]); }
    if (argc >= 13) { kn.adapt_mult = atof(argv[11]); kn.adapt_min = atoi(argv[12]); }
    FILE* log = nullptr;
    if (argc >= 15) { opt.rescue_long_indels = atoi(argv[13]); log = fopen(argv[14], "w"); if (!log) return 2; }
    const rescue::DalignSpec dspec = rescue::spec_for_error(opt.error);
    cns::Rescuer rescuer(dspec, opt.error);
    uint64_t n_tried = 0, n_rescued = 0;
    ora_cns_options oo = {opt.min_align_size, opt.min_cov, opt.max_cov, opt.error, opt.mapping_ratio, opt.use_fixed_ident_cutoff};
    ora_volume reads;
    if (ora_volumes_merge(argv[1], &reads)) { fprintf(stderr, "cannot load %s\n", argv[1]); return 2; }
    std::vector<uint64_t> seq_off(reads.nseq + 1, 0);
    for (uint64_t i = 0; i < reads.nseq; ++i) seq_off[i + 1] = seq_off[i] + reads.size[i];
    int np = 0;
    { FILE* f = fopen((std::string(argv[2]) + ".partitions").c_str(), "r"); if (!f || fscanf(f, "%d", &np) != 1) return 2; fclose(f); }
    uint64_t mism = 0, n_templates = 0, n_overlaps = 0, n_aligned = 0, n_used = 0, n_rounds = 0;
    ora_aligner* al = ora_aligner_new(opt.error);
    for (int p = 0; p < np; ++p) {
        FILE* f = fopen((std::string(argv[2]) + ".p" + std::to_string(p)).c_str(), "rb");
        if (!f) continue;
        fseek(f, 0, SEEK_END); const size_t n = (size_t)ftell(f) / 28; fseek(f, 0, SEEK_SET);
        std::vector<cns::Packed> recs(n);
        if (fread(recs.data(), 28, n, f) != n) return 2;
        fclose(f);
        // the sequential loop
        std::vector<cns::Packed> copy = recs;
        ora_cns_result want; memset(&want, 0, sizeof want);
        ora_cns_partition(&reads, (uint32_t*)copy.data(), n, &oo, &want);
        // the batched loop
        std::vector<necat_candidate> cands; std::vector<uint64_t> off, n_all;
        if (cns::load_partition(recs, seq_off.data(), reads.nseq, cands, off, n_all)) { printf("bad record\n"); return 1; }
        std::vector<cns::Template> ts(n_all.size());
        for (size_t t = 0; t < ts.size(); ++t) {
            ts[t].c = cands.data() + off[t]; ts[t].c_base = off[t]; ts[t].n = (uint32_t)(off[t + 1] - off[t]); ts[t].n_all = (uint32_t)n_all[t];
            ts[t].tsize = (int)cands[off[t]].ssize;
        }
        std::vector<std::vector<uint8_t>> blocks;
        std::vector<uint8_t> qbuf, tbuf;
        cns::AlignFn fn = [&](const necat_candidate* c, uint64_t m, cns::Aligned* out) -> int {
            blocks.emplace_back();
            std::vector<uint8_t>& blk = blocks.back();
            for (uint64_t i = 0; i < m; ++i) {
                qbuf.resize(c[i].qsize + 1); tbuf.resize(c[i].ssize + 1);
                ora_volume_extract(&reads, (uint64_t)c[i].qid, c[i].qdir, qbuf.data());
                ora_volume_extract(&reads, (uint64_t)c[i].sid, 0, tbuf.data());
                ora_align_result r;
                const int ok = ora_onc_align(al, qbuf.data(), (int)c[i].qoff, (int)c[i].qsize, tbuf.data(), (int)c[i].soff, (int)c[i].ssize,
                                             512, opt.min_align_size, 4, &r);
                out[i].a.ok = ok; out[i].a.qoff = r.qoff; out[i].a.qend = r.qend; out[i].a.toff = r.toff; out[i].a.tend = r.tend;
                out[i].a.align_size = r.align_size; out[i].a.ident_perc = r.ident_perc;
                out[i].block = (uint32_t)blocks.size() - 1; out[i].off = blk.size();
                if (opt.rescue_long_indels && cns::extension_short(c[i], out[i].a)) {
                    ++n_tried;
                    if (rescuer.go(c[i], qbuf.data(), tbuf.data(), opt.min_align_size, &out[i].a)) {
                        ++n_rescued;
                        blk.insert(blk.end(), rescuer.cols.begin(), rescuer.cols.end());
                        continue;
                    }
                }
                if (!ok) continue;
                for (int k = 0; k < r.align_size; ++k) {
                    const char q = r.query_align[k], t = r.target_align[k];
                    blk.push_back(q == '-' ? 2 : (t == '-' ? 1 : (q == t ? 0 : 3)));
                }
            }
            return 0;
        };
        cns::Stats st;
        if (cns::run(ts, opt, kn, fn, &st)) return 2;
        n_aligned += st.n_aligned; n_used += st.n_used; n_rounds += st.n_rounds;
        static const char dec[5] = {'A', 'C', 'G', 'T', '-'};
        if (log) {
            // the log of oracle/cns_ref_harness.c, "full"
            std::string qa, ta;
            for (size_t t = 0; t < ts.size(); ++t) {
                const cns::Template& G = ts[t];
                if (!G.examined) continue;
                for (size_t k = 0; k < G.overlaps.size(); ++k) {
                    const necat_cns_overlap& g = G.overlaps[k];
                    const necat_candidate& c = cands[g.cand];
                    qbuf.resize(c.qsize + 1); tbuf.resize(c.ssize + 1);
                    ora_volume_extract(&reads, (uint64_t)c.qid, c.qdir, qbuf.data());
                    ora_volume_extract(&reads, (uint64_t)c.sid, 0, tbuf.data());
                    const uint8_t* ops = blocks[g.ops_block].data() + g.ops_off;
                    qa.clear(); ta.clear();
                    int q = g.qoff, tt = g.toff;
                    for (int x = 0; x < g.align_size; ++x) {
                        qa.push_back(ops[x] == 2 ? '-' : dec[qbuf[q]]); ta.push_back(ops[x] == 1 ? '-' : dec[tbuf[tt]]);
                        q += ops[x] != 2; tt += ops[x] != 1;
                    }
                    auto fnv = [](const std::string& s) { unsigned long long h = 1469598103934665603ULL; for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ULL; } return h; };
                    fprintf(log, "A\t%d\t%d\t%.17g\t%d\t%016llx\t%016llx\t%s\t%s\n", g.toff, g.tend, g.weight, g.align_size, fnv(qa), fnv(ta), qa.c_str(), ta.c_str());
                    ++n_overlaps;
                }
                fprintf(log, "T\t%d\t%d\t%.17g\t%d\t%d\t%zu", G.c[0].sid, G.tsize, G.ident_cutoff, G.num_can, G.num_ovlps, (size_t)G.ranges.size() / 2);
                for (size_t k = 0; k < G.ranges.size(); ++k) fprintf(log, "\t%d", G.ranges[k]);
                fprintf(log, "\n");
                ++n_templates;
            }
        }
        if (opt.rescue_long_indels) { ora_cns_result_free(&want); continue; }
        // compare
        if (want.n_templates != ts.size()) { printf("partition %d: %zu templates, oracle %zu\n", p, ts.size(), want.n_templates); ++mism; continue; }
        for (size_t t = 0; t < ts.size(); ++t) {
            const ora_cns_template& W = want.templates[t];
            const cns::Template& G = ts[t];
            ++n_templates;
            bool bad = W.examined != (int)G.examined || W.template_id != G.c[0].sid;
            if (!bad && W.examined) {
                bad = W.ident_cutoff != G.ident_cutoff || W.num_can != G.num_can || W.num_ovlps != G.num_ovlps ||
                      W.ovlp_end - W.ovlp_begin != G.overlaps.size() || 2 * (W.range_end - W.range_begin) != G.ranges.size();
                for (size_t k = 0; !bad && k < G.ranges.size(); ++k) bad = want.ranges[2 * W.range_begin + k] != G.ranges[k];
                for (size_t k = 0; !bad && k < G.overlaps.size(); ++k) {
                    const ora_cns_overlap& w = want.overlaps[W.ovlp_begin + k];
                    const necat_cns_overlap& g = G.overlaps[k];
                    bad = (uint64_t)w.cand != g.cand - G.c_base || w.qoff != g.qoff || w.qend != g.qend || w.toff != g.toff || w.tend != g.tend ||
                          w.align_size != g.align_size || w.ident_perc != g.ident_perc || w.weight != g.weight;
                    if (bad) break;
                    const necat_candidate& c = cands[g.cand];
                    qbuf.resize(c.qsize + 1); tbuf.resize(c.ssize + 1);
                    ora_volume_extract(&reads, (uint64_t)c.qid, c.qdir, qbuf.data());
                    ora_volume_extract(&reads, (uint64_t)c.sid, 0, tbuf.data());
                    const uint8_t* ops = blocks[g.ops_block].data() + g.ops_off;
                    int q = g.qoff, tt = g.toff;
                    for (int x = 0; x < g.align_size && !bad; ++x) {
                        const char qc = ops[x] == 2 ? '-' : dec[qbuf[q]], tc = ops[x] == 1 ? '-' : dec[tbuf[tt]];
                        bad = qc != want.strs[w.str_at + x] || tc != want.strs[w.str_at + w.align_size + x];
                        q += ops[x] != 2; tt += ops[x] != 1;
                    }
                    ++n_overlaps;
                }
            }
            if (bad) {
                if (mism < 5) printf("template %d: cutoff %.17g/%.17g num_can %d/%d ovlps %d/%d\n", W.template_id, W.ident_cutoff, G.ident_cutoff,
                                     W.num_can, G.num_can, W.num_ovlps, G.num_ovlps);
                ++mism;
            }
        }
        ora_cns_result_free(&want);
    }
    if (log) fclose(log);
    printf("cns_mismatch=%lu templates=%lu overlaps=%lu aligned=%lu used=%lu rounds=%lu rescue_tried=%lu rescued=%lu\n", (unsigned long)mism,
           (unsigned long)n_templates, (unsigned long)n_overlaps, (unsigned long)n_aligned, (unsigned long)n_used, (unsigned long)n_rounds,
           (unsigned long)n_tried, (unsigned long)n_rescued);
    return mism ? 1 : 0;
}
