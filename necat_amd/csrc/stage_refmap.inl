// stage_refmap.inl - reads against a reference (oc2rm_worker).
// One of the stage files of libnecat_hip.so's single translation unit: necat_hip.hip includes them in order, inside its extern "C" block, after the
// context / knob / result-pool code they all use (the kernels are header templates and the stages share host helpers: one device code object, one 30 s build).

// ------------------------------------------------------------------------------------------ reads against a reference (oc2rm_worker)

int necat_map_reference(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                        int read_start_id, int ref_start_id, const necat_map_options* opt,
                        necat_m4** out, uint64_t* n_out, uint64_t* n_candidates, uint64_t* n_rescued)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ix || !ref || !reads || !opt || !out || !n_out) return NECAT_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (n_candidates) *n_candidates = 0;
    if (n_rescued) *n_rescued = 0;
    necat_map_options o = *opt;
    o.job = 1;                                   // rm_worker.c:251-252: sorted, cut to num_candidates
    DevCands dev;
    int rc = find_impl(ctx, ix, ref, reads, read_start_id, ref_start_id, 0 /* pairwise = FALSE, rm_worker.c:231 */, &o, nullptr, nullptr, &dev);
    if (rc) return rc;
    if (n_candidates) *n_candidates = dev.n;
    if (dev.n == 0) { *out = (necat_m4*)result_alloc(sizeof(necat_m4)); return *out ? NECAT_OK : set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"); }
    RmOut ro;
    necat_m4* unused = nullptr; uint64_t unused_n = 0;
    if ((rc = extend_impl(ctx, ref, reads, read_start_id, ref_start_id, nullptr, dev.n, &o, 1 /* ONC_TAIL_MATCH_LEN_SHORT, rm_worker.c:92 */,
                          &unused, &unused_n, nullptr, &dev, nullptr, &ro))) return rc;
    const double w0 = wall_ms();
    // the bases come back to the host only if some candidate needs the rescue pair
    const uint64_t ng = ro.group_off.empty() ? 0 : ro.group_off.size() - 1;
    bool any = false;
    for (uint64_t i = 0; i < dev.n && !any; ++i) any = ro.ok[i] && rm::needs_rescue(ro.cands[i], ro.m4[i]);
    std::vector<u64> w_reads, w_ref;
    if (any) {
        w_reads.resize((reads->nbases + 31) / 32 + 1); w_ref.resize((ref->nbases + 31) / 32 + 1);
        NECAT_HIP(ctx, hipMemcpy(w_reads.data(), reads->bases, (w_reads.size() - 1) * 8, hipMemcpyDeviceToHost));
        NECAT_HIP(ctx, hipMemcpy(w_ref.data(), ref->bases, (w_ref.size() - 1) * 8, hipMemcpyDeviceToHost));
    }
    rm::Words hr, hf;
    hr.w = w_reads.data(); hr.seq_off = reads->h_seq_off.data();
    hf.w = w_ref.data(); hf.seq_off = ref->h_seq_off.data();
    const rescue::DalignSpec dspec = rescue::spec_for_error(o.error);
    // groups (reads) are dealt out in runs of 16; every run's records are kept apart and joined in read order
    const uint64_t run = 16, nruns = (ng + run - 1) / run;
    std::vector<std::vector<necat_m4>> parts(nruns);
    std::atomic<uint64_t> next(0), tried(0), rescued(0);
    auto work = [&]() {
        rm::Worker wk(dspec, o.error);
        for (;;) {
            const uint64_t r = next.fetch_add(1);
            if (r >= nruns) break;
            for (uint64_t g = r * run; g < std::min(ng, (r + 1) * run); ++g)
                wk.replay(ro.cands.data(), ro.m4.data(), ro.ok.data(), ro.group_off[g], ro.group_off[g + 1], hr, hf, read_start_id, ref_start_id,
                          o.align_size_cutoff, parts[r]);
        }
        tried += wk.n_rescue_tried; rescued += wk.n_rescued;
    };
    unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 32u));
    nt = (unsigned)std::min<uint64_t>(nt, std::max<uint64_t>(1, nruns));
    std::vector<std::thread> th;
    for (unsigned x = 0; x + 1 < nt; ++x) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    uint64_t total = 0;
    for (auto& p : parts) total += p.size();
    necat_m4* res = (necat_m4*)result_alloc(std::max<uint64_t>(1, total) * sizeof(necat_m4));
    if (!res) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
    uint64_t at = 0;
    for (auto& p : parts) { if (!p.empty()) memcpy(res + at, p.data(), p.size() * sizeof(necat_m4)); at += p.size(); }
    if (n_rescued) *n_rescued = rescued.load();
    if (g_trace & 2) fprintf(stderr, "[necat] map_reference host: %.2f ms, %lu candidates, %lu rescue attempts, %lu rescued, %lu records\n", wall_ms() - w0,
                             (unsigned long)dev.n, (unsigned long)tried.load(), (unsigned long)rescued.load(), (unsigned long)total);
    *out = res; *n_out = total;
    return NECAT_OK;
}
