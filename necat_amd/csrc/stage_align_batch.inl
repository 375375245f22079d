// stage_align_batch.inl - alignments with their columns (necat_onc_align_batch, necat_gapped_strings).
// One of the stage files of libnecat_hip.so's single translation unit: necat_hip.hip includes them in order, inside its extern "C" block, after the
// context / knob / result-pool code they all use (the kernels are header templates and the stages share host helpers: one device code object, one 30 s build).

// ------------------------------------------------------------------------------------------ alignments with their columns (the consensus stage's call, necat_onc_align_batch)
int necat_onc_align_batch(necat_ctx* ctx, const necat_volume* ref, const necat_volume* reads, int read_start_id, int ref_start_id,
                          const necat_candidate* cands, uint64_t n, const necat_map_options* opt, int tail_match_len,
                          necat_alignment** aln, uint8_t** ops, uint64_t** ops_off)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ref || !reads || !opt || !aln || !ops || !ops_off || (n && !cands)) return NECAT_ERR_ARG;
    *aln = nullptr; *ops = nullptr; *ops_off = nullptr;
    AlignOut ao;
    ao.aln = (necat_alignment*)result_alloc(std::max<uint64_t>(1, n) * sizeof(necat_alignment));
    if (!ao.aln) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
    ao.off.assign(n + 1, 0);
    if (n) {
        const int rc = extend_impl(ctx, ref, reads, read_start_id, ref_start_id, cands, n, opt, tail_match_len, nullptr, nullptr, &ao);
        if (rc) { necat_free(ao.aln); for (auto& pr : ao.parts) necat_free(pr.first); return rc; }
    }
    uint64_t* f = (uint64_t*)result_alloc((n + 1) * 8);
    uint8_t* o = nullptr;
    if (ao.parts.size() == 1) { o = ao.parts[0].first; ao.parts.clear(); }       // the usual case: one batch, no copy
    else {
        o = (uint8_t*)result_alloc(std::max<uint64_t>(1, ao.total));
        uint64_t at = 0;
        if (o) for (auto& pr : ao.parts) { memcpy(o + at, pr.first, pr.second); at += pr.second; }
        for (auto& pr : ao.parts) necat_free(pr.first);
        ao.parts.clear();
    }
    if (!o || !f) { necat_free(ao.aln); necat_free(o); necat_free(f); return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"); }
    memcpy(f, ao.off.data(), (n + 1) * 8);
    *aln = ao.aln; *ops = o; *ops_off = f;
    return NECAT_OK;
}

int necat_gapped_strings(const uint8_t* ops, uint64_t n, const uint8_t* qseq, uint64_t qsize, uint64_t qoff,
                         const uint8_t* tseq, uint64_t tsize, uint64_t toff, char* query_align, char* target_align)
{
    if ((n && (!ops || !query_align || !target_align)) || !qseq || !tseq) return NECAT_ERR_ARG;
    static const char dec[5] = {'A', 'C', 'G', 'T', '-'};      // DecodeDNA / GAP_CHAR (common/ontcns_defs.h:36-39)
    uint64_t q = qoff, t = toff;
    for (uint64_t i = 0; i < n; ++i) {
        const int op = (ops[i >> 2] >> ((i & 3) * 2)) & 3;
        if ((op != 2 && q >= qsize) || (op != 1 && t >= tsize)) return NECAT_ERR_ARG;
        query_align[i] = op == 2 ? '-' : dec[qseq[q] & 3];
        target_align[i] = op == 1 ? '-' : dec[tseq[t] & 3];
        q += op != 2; t += op != 1;
    }
    return NECAT_OK;
}
