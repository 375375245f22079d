// stage_cns.inl - the consensus stage's extension loop (cns_loop.h).
// One of the stage files of libnecat_hip.so's single translation unit: necat_hip.hip includes them in order, inside its extern "C" block, after the
// context / knob / result-pool code they all use (the kernels are header templates and the stages share host helpers: one device code object, one 30 s build).

// ------------------------------------------------------------------------------------------ consensus stage: the extension loop

void necat_cns_default_options(necat_cns_options* o)
{   // consensus/cns_options.c:10-22
    o->min_align_size = 400; o->min_cov = 4; o->max_cov = 12; o->error = 0.5; o->mapping_ratio = 0.8; o->use_fixed_ident_cutoff = 0;
    o->rescue_long_indels = 0;
}

int necat_cns_load_partition(necat_ctx* ctx, const necat_volume* reads, const void* packed, uint64_t n,
                             necat_candidate** cands, uint64_t** tmpl_off, uint64_t** n_all, uint64_t* n_templates)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !reads || (n && !packed) || !cands || !tmpl_off || !n_all || !n_templates) return NECAT_ERR_ARG;
    *cands = nullptr; *tmpl_off = nullptr; *n_all = nullptr; *n_templates = 0;
    std::vector<cns::Packed> recs(n);
    if (n) memcpy(recs.data(), packed, n * sizeof(cns::Packed));
    std::vector<necat_candidate> c; std::vector<uint64_t> off, na;
    const uint64_t bad = cns::load_partition(recs, reads->h_seq_off.data(), reads->nseq, c, off, na);
    if (bad) return set_err(ctx, NECAT_ERR_ARG, "candidate record %lu refers to a read outside the read set or has a range outside its reads", (unsigned long)(bad - 1));
    necat_candidate* oc = (necat_candidate*)malloc(std::max<size_t>(1, c.size()) * sizeof(necat_candidate));
    uint64_t* oo = (uint64_t*)malloc(off.size() * 8);
    uint64_t* on = (uint64_t*)malloc(std::max<size_t>(1, na.size()) * 8);
    if (!oc || !oo || !on) { free(oc); free(oo); free(on); return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"); }
    if (!c.empty()) memcpy(oc, c.data(), c.size() * sizeof(necat_candidate));
    memcpy(oo, off.data(), off.size() * 8);
    if (!na.empty()) memcpy(on, na.data(), na.size() * 8);
    *cands = oc; *tmpl_off = oo; *n_all = on; *n_templates = na.size();
    return NECAT_OK;
}

void necat_cns_result_free(necat_cns_result* r)
{
    if (!r) return;
    for (uint32_t b = 0; b < r->n_ops_blocks; ++b) necat_free(r->ops[b]);
    free(r->ops); free(r->templates); free(r->overlaps); free(r->ranges);
    free(r);
}

int necat_cns_extension_batch(necat_ctx* ctx, const necat_volume* reads, const necat_candidate* cands, const uint64_t* tmpl_off,
                              const uint64_t* n_all, uint64_t n_templates, const necat_cns_options* opt, necat_cns_result** out)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !reads || !opt || !out || (n_templates && (!tmpl_off || !cands))) return NECAT_ERR_ARG;
    *out = nullptr;
    if (opt->max_cov < 1 || opt->max_cov > 60000 || opt->min_align_size < 0 || !(opt->error > 0.0 && opt->error <= 1.0))
        return set_err(ctx, NECAT_ERR_ARG, "consensus options out of range");
    const double w0 = wall_ms();
    std::vector<cns::Template> ts(n_templates);
    for (uint64_t t = 0; t < n_templates; ++t) {
        const uint64_t lo = tmpl_off[t], hi = tmpl_off[t + 1];
        if (hi < lo || hi - lo >= (1ULL << 31)) return set_err(ctx, NECAT_ERR_ARG, "template %lu: bad candidate range", (unsigned long)t);
        cns::Template& T = ts[t];
        T.c = cands + lo; T.c_base = lo; T.n = (uint32_t)(hi - lo); T.n_all = n_all ? (uint32_t)std::min<uint64_t>(n_all[t], 0xffffffffu) : T.n;
        for (uint64_t i = lo; i < hi; ++i) {
            const necat_candidate& c = cands[i];
            if (c.sid != cands[lo].sid || c.sdir != 0 || c.sid < 0 || (uint64_t)c.sid >= reads->nseq || c.qid < 0 || (uint64_t)c.qid >= reads->nseq ||
                c.ssize != reads->h_seq_off[c.sid + 1] - reads->h_seq_off[c.sid] || c.qsize != reads->h_seq_off[c.qid + 1] - reads->h_seq_off[c.qid] ||
                c.sbeg > c.send || c.send > c.ssize || c.qoff > c.qsize || c.soff > c.ssize || c.ssize >= (1ULL << 31) || c.qsize >= (1ULL << 31))
                return set_err(ctx, NECAT_ERR_ARG, "candidate %lu of template %lu is inconsistent (one forward subject per template, ranges inside the reads)",
                               (unsigned long)(i - lo), (unsigned long)t);
        }
        T.tsize = T.n ? (int)cands[lo].ssize : 0;
    }
    necat_map_options mo; necat_default_options(&mo);
    mo.error = opt->error; mo.align_size_cutoff = opt->min_align_size;
    std::vector<u8*> blocks;
    double device_ms = 0, align_wall = 0;
    // -r 1: the host pair (cns_rescue.h) on the candidates of a pass whose block-wise extension failed or fell short.  The reads
    // come back from the device once per call (2-bit words, base i in bits 2 (i & 31) of word i >> 5).
    std::vector<u64> h_words;
    const rescue::DalignSpec dspec = opt->rescue_long_indels ? rescue::spec_for_error(opt->error) : rescue::DalignSpec();
    uint64_t n_rescue_tried = 0, n_rescued = 0;
    double rescue_ms = 0;
    auto rescue_pass = [&](const necat_candidate* c, uint64_t m, cns::Aligned* res) -> int {
        const double r0 = wall_ms();
        std::vector<uint64_t> need;
        for (uint64_t i = 0; i < m; ++i) if (cns::extension_short(c[i], res[i].a)) need.push_back(i);
        if (need.empty()) return NECAT_OK;
        if (h_words.empty()) {
            h_words.resize((reads->nbases + 31) / 32 + 1);
            NECAT_HIP(ctx, hipMemcpy(h_words.data(), reads->bases, (h_words.size() - 1) * 8, hipMemcpyDeviceToHost));
        }
        struct Got { bool ok = false; necat_alignment a; std::vector<u8> packed; };
        std::vector<Got> got(need.size());
        std::atomic<size_t> next(0);
        auto work = [&]() {
            cns::Rescuer rs(dspec, opt->error);
            std::vector<u8> q, t;
            auto decode = [&](int32_t id, int rev, std::vector<u8>& dst) {
                const u64 b = reads->h_seq_off[id], n = reads->h_seq_off[id + 1] - b;
                dst.resize(n);
                if (!rev) for (u64 i = 0; i < n; ++i) dst[i] = (u8)((h_words[(b + i) >> 5] >> (((b + i) & 31) * 2)) & 3);
                else for (u64 i = 0; i < n; ++i) { const u64 g = b + n - 1 - i; dst[i] = (u8)(3 - ((h_words[g >> 5] >> ((g & 31) * 2)) & 3)); }
            };
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= need.size()) break;
                const necat_candidate& cc = c[need[k]];
                decode(cc.qid, cc.qdir, q); decode(cc.sid, 0, t);
                Got& g = got[k];
                g.a = res[need[k]].a;
                g.ok = rs.go(cc, q.data(), t.data(), opt->min_align_size, &g.a);
                if (!g.ok) continue;
                g.packed.assign((rs.cols.size() + 3) / 4, 0);
                for (size_t j = 0; j < rs.cols.size(); ++j) g.packed[j >> 2] |= (u8)(rs.cols[j] << (2 * (j & 3)));
            }
        };
        unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 32u));
        nt = (unsigned)std::min<size_t>(nt, need.size());
        std::vector<std::thread> th;
        for (unsigned x = 0; x + 1 < nt; ++x) th.emplace_back(work);
        work();
        for (auto& x : th) x.join();
        u64 bytes = 0;
        for (const Got& g : got) if (g.ok) bytes += (g.packed.size() + 7) & ~(u64)7;
        n_rescue_tried += need.size();
        if (bytes) {
            u8* blk = (u8*)result_alloc(bytes);
            if (!blk) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
            const u32 bi = (u32)blocks.size();
            blocks.push_back(blk);
            u64 at = 0;
            for (size_t k = 0; k < got.size(); ++k) {
                const Got& g = got[k];
                if (!g.ok) continue;
                memcpy(blk + at, g.packed.data(), g.packed.size());
                res[need[k]].a = g.a; res[need[k]].block = bi; res[need[k]].off = at;
                at += (g.packed.size() + 7) & ~(u64)7;
                ++n_rescued;
            }
        }
        rescue_ms += wall_ms() - r0;
        if (g_trace & 2) fprintf(stderr, "[necat] cns rescue: %zu of %lu candidates tried, %.2f ms\n", need.size(), (unsigned long)m, wall_ms() - r0);
        return NECAT_OK;
    };
    cns::AlignFn fn = [&](const necat_candidate* c, uint64_t m, cns::Aligned* res) -> int {
        const double a0 = wall_ms();
        AlignOut ao;
        ao.defer_copy = true;       // the loop only needs the coordinates to go on; the columns arrive while it does
        ao.aln = (necat_alignment*)result_alloc(std::max<uint64_t>(1, m) * sizeof(necat_alignment));
        if (!ao.aln) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
        ao.off.assign(m + 1, 0);
        const int rc = extend_impl(ctx, reads, reads, 0, 0, c, m, &mo, 4 /* ONC_TAIL_MATCH_LEN_LONG, oc_aligner.h:42 */, nullptr, nullptr, &ao);
        if (rc) {
            if (ctx->stream_copy) (void)hipStreamSynchronize(ctx->stream_copy);
            ctx->copy_pending = false;
            necat_free(ao.aln); for (auto& pr : ao.parts) necat_free(pr.first); return rc;
        }
        device_ms += ctx->tm.extend_ms;
        // the columns stay where the device copied them: one block per batch of the pass
        size_t p = 0; u64 p_start = 0;
        const u32 b0 = (u32)blocks.size();
        for (auto& pr : ao.parts) blocks.push_back(pr.first);
        for (uint64_t i = 0; i < m; ++i) {
            res[i].a = ao.aln[i];
            const u64 at = ao.off[i];
            while (p < ao.parts.size() && at >= p_start + ao.parts[p].second && ao.off[i + 1] > at) { p_start += ao.parts[p].second; ++p; }
            res[i].block = b0 + (u32)std::min(p, ao.parts.empty() ? 0 : ao.parts.size() - 1);
            res[i].off = at - p_start;
        }
        necat_free(ao.aln);
        align_wall += wall_ms() - a0;
        if (g_trace & 2) fprintf(stderr, "[necat] cns pass: %lu alignments, %.2f ms\n", (unsigned long)m, wall_ms() - a0);
        return opt->rescue_long_indels ? rescue_pass(c, m, res) : NECAT_OK;
    };
    cns::Knobs kn; kn.spec_estimate_extra = g_cns_spec_extra; kn.spec_cover = g_cns_spec_cover;
    cns::Stats st;
    const double w_run = wall_ms();
    if (!ctx->cns_scratch) ctx->cns_scratch = new cns::Scratch();
    const int rc = cns::run(ts, *opt, kn, fn, &st, (cns::Scratch*)ctx->cns_scratch);
    if (g_trace & 2) fprintf(stderr, "[necat] cns host: setup %.2f ms, init %.2f, select %.2f, gather %.2f, replay %.2f ms\n", w_run - w0, st.init_ms, st.select_ms,
                             st.gather_ms, st.replay_ms);
    auto drop = [&]() { for (u8* b : blocks) necat_free(b); };
    {   // the last columns may still be on their way
        const hipError_t e = ctx->stream_copy ? hipStreamSynchronize(ctx->stream_copy) : hipSuccess;
        ctx->copy_pending = false;
        if (e != hipSuccess && !rc) { drop(); return set_err(ctx, NECAT_ERR_DEVICE, "column copy failed: %s", hipGetErrorString(e)); }
    }
    if (rc) { drop(); return rc; }
    necat_cns_result* r = (necat_cns_result*)calloc(1, sizeof(necat_cns_result));
    uint64_t n_ov = 0, n_rg = 0;
    for (auto& T : ts) { n_ov += T.overlaps.size(); n_rg += T.ranges.size() / 2; }
    if (r) {
        r->templates = (necat_cns_template*)calloc(std::max<uint64_t>(1, n_templates), sizeof(necat_cns_template));
        r->overlaps = (necat_cns_overlap*)malloc(std::max<uint64_t>(1, n_ov) * sizeof(necat_cns_overlap));
        r->ranges = (int32_t*)malloc(std::max<uint64_t>(1, n_rg) * 8);
        r->ops = (uint8_t**)malloc(std::max<size_t>(1, blocks.size()) * sizeof(uint8_t*));
    }
    if (!r || !r->templates || !r->overlaps || !r->ranges || !r->ops) {
        drop();
        if (r) { free(r->templates); free(r->overlaps); free(r->ranges); free(r->ops); free(r); }
        return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
    }
    {
        uint64_t ov = 0, rg = 0;
        for (uint64_t t = 0; t < n_templates; ++t) {
            necat_cns_template& o = r->templates[t];
            o.ovlp_begin = ov; o.range_begin = rg;
            ov += ts[t].overlaps.size(); rg += ts[t].ranges.size() / 2;
            o.ovlp_end = ov; o.range_end = rg;
        }
    }
    cns::parallel_for(n_templates, [&](size_t t) {
        const cns::Template& T = ts[t];
        necat_cns_template& o = r->templates[t];
        o.examined = T.examined ? 1 : 0; o.num_can = T.num_can; o.num_ovlps = T.num_ovlps; o.ident_cutoff = T.ident_cutoff;
        if (!T.overlaps.empty()) memcpy(r->overlaps + o.ovlp_begin, T.overlaps.data(), T.overlaps.size() * sizeof(necat_cns_overlap));
        if (!T.ranges.empty()) memcpy(r->ranges + 2 * o.range_begin, T.ranges.data(), T.ranges.size() * 4);
    });
    r->n_templates = n_templates; r->n_overlaps = n_ov; r->n_ranges = n_rg;
    r->n_ops_blocks = (uint32_t)blocks.size();
    for (size_t b = 0; b < blocks.size(); ++b) r->ops[b] = blocks[b];
    r->n_aligned = st.n_aligned; r->n_used = st.n_used; r->n_rounds = st.n_rounds;
    r->device_ms = device_ms; r->host_ms = wall_ms() - w0 - align_wall - rescue_ms;
    r->n_rescue_tried = n_rescue_tried; r->n_rescued = n_rescued; r->rescue_ms = rescue_ms;
    if (g_trace & 2) fprintf(stderr, "[necat] cns total %.2f ms: passes %.2f (device events %.2f), host %.2f\n", wall_ms() - w0, align_wall, device_ms, r->host_ms);
    ctx->tm.extend_ms = device_ms;
    *out = r;
    return NECAT_OK;
}
