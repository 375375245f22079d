// stage_asm_plan.inl - oc2asmpm: votes and chained ranges on the device (asm_plan.h).
// One of the stage files of libnecat_hip.so's single translation unit: necat_hip.hip includes them in order, inside its extern "C" block, after the
// context / knob / result-pool code they all use (the kernels are header templates and the stages share host helpers: one device code object, one 30 s build).

// ------------------------------------------------------------------------------------------ oc2asmpm: votes and chained ranges on the device (asm_plan.h)

int necat_asm_plan_batch(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads, int read_start_id, int ref_start_id,
                         const necat_map_options* opt, necat_asm_plan** out, uint64_t** first)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ix || !ref || !reads || !opt || !out || !first) return NECAT_ERR_ARG;
    *out = nullptr; *first = nullptr;
    if (opt->kmer_size != ix->k) return set_err(ctx, NECAT_ERR_ARG, "index was built for k=%d, options say %d", ix->k, opt->kmer_size);
    if (opt->scan_window < 1 || opt->num_candidates < 1 || opt->num_candidates > 65536) return set_err(ctx, NECAT_ERR_ARG, "scan_window / num_candidates out of range");
    if (ref->nbases >= (1ULL << 31) || reads->nbases >= (1ULL << 31)) return set_err(ctx, NECAT_ERR_ARG, "volume too large for 32-bit offsets (asm_pm_common.c keeps them in int)");
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const u32 nreads = (u32)reads->nseq;
    const int NE = opt->num_candidates;
    uint64_t* fo = (uint64_t*)result_alloc(((size_t)nreads + 1) * 8);
    if (!fo) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
    auto fail = [&](int rc) { necat_free(fo); return rc; };
    if (nreads == 0) { fo[0] = 0; *first = fo; *out = (necat_asm_plan*)result_alloc(sizeof(necat_asm_plan)); return NECAT_OK; }
    const DevVolume dref = dev_view(ref), drd = dev_view(reads);
    int rc;
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_prev = t_begin;
    auto tick = [&](const char* what) {          // (host clock between the calls' own synchronisation points; NECAT_TRACE=4 - it must not add any: the chunks overlap)
        if (!(g_trace & 4)) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[necat] asm plan %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // ---- hit counts per read-strand (k_seed_hits with z = BC), the table words kept
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_MISC], (size_t)nreads * 8 + 128)) ||
        (rc = buf_ensure(ctx, ctx->scratch[SC_SEED_KST], 2 * (reads->nbases / (u64)opt->scan_window + nreads + 2) * 8))) return fail(rc);
    u32* d_hits = (u32*)ctx->scratch[SC_MISC].p;
    int* d_err = (int*)((char*)ctx->scratch[SC_MISC].p + (((size_t)nreads * 8 + 63) & ~(size_t)63));
    u64* d_kst = (u64*)ctx->scratch[SC_SEED_KST].p;
    hipLaunchKernelGGL(k_seed_hits, dim3(grid_for((u64)nreads * 64, 256)), dim3(256), 0, s, drd, index_view(ix), opt->kmer_size, opt->scan_window, 0u, nreads, d_hits, d_kst);
    if (hipGetLastError() != hipSuccess) return fail(set_err(ctx, NECAT_ERR_DEVICE, "k_seed_hits launch failed"));
    std::vector<u32> hits((size_t)nreads * 2);
    if (hipMemsetAsync(d_err, 0, 4, s) != hipSuccess || hipMemcpyAsync(hits.data(), d_hits, (size_t)nreads * 8, hipMemcpyDeviceToHost, s) != hipSuccess)
        return fail(set_err(ctx, NECAT_ERR_DEVICE, "asm plan: hit counts"));
    // ---- per (subject, strand): occurrences of the sampled 10-mers (beside the copy above)
    u64 ref_lmax = 0;
    for (u64 q = 0; q < ref->nseq; ++q) ref_lmax = std::max(ref_lmax, ref->h_seq_off[q + 1] - ref->h_seq_off[q]);
    u32 cap_max = 64; while (cap_max < 2 * (ref_lmax / kAsmRangeW + 1)) cap_max <<= 1;
    const u32 occ_waves = (u32)std::max<u64>(1, std::min<u64>(std::min<u64>(2 * ref->nseq, 4096), ((u64)1 << 30) / ((u64)cap_max * 8)));
    const size_t occ_bytes = 2 * (ref->nbases / kAsmRangeW + ref->nseq + 2);
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_ASM_OCC], occ_bytes)) || (rc = buf_ensure(ctx, ctx->scratch[SC_ASM_TAB], (size_t)occ_waves * cap_max * 8))) return fail(rc);
    u8* d_occ = (u8*)ctx->scratch[SC_ASM_OCC].p;
    if (hipMemsetAsync(ctx->scratch[SC_ASM_TAB].p, 0, (size_t)occ_waves * cap_max * 8, s) != hipSuccess) return fail(set_err(ctx, NECAT_ERR_DEVICE, "asm plan: memset"));
    if (ref->nseq) hipLaunchKernelGGL(k_asm_subj_occ, dim3(occ_waves), dim3(64), 0, s, dref, (u32*)ctx->scratch[SC_ASM_TAB].p, cap_max, d_occ);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return fail(set_err(ctx, NECAT_ERR_DEVICE, "k_asm_subj_occ failed: %s", hipGetErrorString(hipGetLastError())));
    tick("hits + subject occurrences");
    // ---- reads in descending work order, chunks bounded by a scratch budget (the pool of 384-byte blocks is sized by the hit counts)
    std::vector<u32> order(nreads);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return std::max(hits[2 * (size_t)a], hits[2 * (size_t)a + 1]) > std::max(hits[2 * (size_t)b], hits[2 * (size_t)b + 1]); });
    // (the pool of 384-byte blocks is sized by the hit counts, an upper bound several times the blocks really touched: the budget is what keeps a
    // chunk inside HBM.  A chunk's vote kernels are as long as the walk of its heaviest read, so the chunks run on TWO arena sets and two streams:
    // chunk i + 1's vote kernels are in flight while chunk i's tail finishes and its range stage runs.  16 M blocks = 6 GB per set by default - a
    // short-lived process pays for the device memory it maps (the first 30 GB of arenas of a process on a fresh box took 0.9 s,
    // profiles/NOTES_r04.md 4) - and the two sets' pools + candidate lists (the per-block bytes below: both scale with the budget; the hash tables, the
    // selection and read-index arenas are small beside them) together never more than 40 % of the memory that is free now)
    u64 budget_blocks = getenv("NECAT_ASM_VOTE_BUDGET") ? std::max<u64>(1024, strtoull(getenv("NECAT_ASM_VOTE_BUDGET"), nullptr, 10)) : (u64)16 << 20;
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr) budget_blocks = std::max<u64>(1 << 16, std::min<u64>(budget_blocks, (u64)(fr * 0.4) / (2 * (sizeof(VBlock) + sizeof(VoteCand)))));
    }
    static const u64 budget_seeds = getenv("NECAT_ASM_SEED_BUDGET") ? std::max<u64>(1024, strtoull(getenv("NECAT_ASM_SEED_BUDGET"), nullptr, 10)) : (u64)32 << 20;
    static const bool overlap = !getenv("NECAT_ASM_NO_OVERLAP");            // (A/B: one arena set, one stream, chunk after chunk)
    VoteParams P; P.k = opt->kmer_size; P.bc = opt->scan_window; P.read_start_id = read_start_id; P.ref_start_id = ref_start_id; P.num_extended = NE;
    std::vector<std::vector<necat_asm_plan>> per_read(nreads);
    u64 tot_pairs = 0, tot_seeds = 0, tot_plans = 0;
    // the chunks, and every per-chunk arena sized once for the largest of them (a grow-only arena that grows chunk by chunk is freed and
    // allocated again each time)
    auto both = [&](u32 r) { return (u64)hits[2 * (size_t)r] + hits[2 * (size_t)r + 1] + 2; };
    std::vector<u32> chunk_end;
    static const ScratchId kSet[2][7] = {{SC_ASM_VMETA, SC_ASM_VHT, SC_ASM_VPOOL, SC_ASM_VOUT, SC_ASM_SEL, SC_ASM_RIDX, SC_ASM_RNEXT},
                                         {SC_ASM_VMETA2, SC_ASM_VHT2, SC_ASM_VPOOL2, SC_ASM_VOUT2, SC_ASM_SEL2, SC_ASM_RIDX2, SC_ASM_RNEXT2}};
    {
        u64 mx_n = 0, mx_ht = 0, mx_pool = 0, mx_tab = 0, mx_next = 0;
        for (u32 p0 = 0; p0 < nreads;) {
            u64 acc = 0, ht = 0, tab = 0, nx = 0; u32 h1 = p0;
            while (h1 < nreads && (h1 == p0 || acc + both(order[h1]) <= budget_blocks)) {
                const u32 r = order[h1];
                acc += both(r);
                for (int st = 0; st < 2; ++st) { const u64 H = std::max<u64>(1, hits[2 * (size_t)r + st]); u64 cap = 4; while (cap < 2 * H) cap <<= 1; ht += cap; }
                const u64 L = reads->h_seq_off[r + 1] - reads->h_seq_off[r];
                tab += 2 * (L + L / 2 + 64); nx += L + 1;
                ++h1;
            }
            mx_n = std::max<u64>(mx_n, h1 - p0); mx_ht = std::max(mx_ht, ht); mx_pool = std::max(mx_pool, acc); mx_tab = std::max(mx_tab, tab); mx_next = std::max(mx_next, nx);
            chunk_end.push_back(h1);
            p0 = h1;
        }
        const size_t meta_bytes = (size_t)mx_n * (sizeof(VoteMeta) + sizeof(ReadIdxMeta) + 4 /* order */ + 8 /* nblk */ + 8 /* nstrand */ + 4 /* nplan */) + 512;
        for (int e = 0; e < ((overlap && chunk_end.size() > 1) ? 2 : 1); ++e)
            if ((rc = buf_ensure(ctx, ctx->scratch[kSet[e][0]], meta_bytes)) ||
                (rc = buf_ensure(ctx, ctx->scratch[kSet[e][1]], mx_ht * 8)) ||
                (rc = buf_ensure(ctx, ctx->scratch[kSet[e][2]], mx_pool * sizeof(VBlock))) ||
                (rc = buf_ensure(ctx, ctx->scratch[kSet[e][3]], mx_pool * sizeof(VoteCand))) ||
                (rc = buf_ensure(ctx, ctx->scratch[kSet[e][4]], (size_t)mx_n * NE * (sizeof(VoteCand) + sizeof(AsmPlanDev)))) ||
                (rc = buf_ensure(ctx, ctx->scratch[kSet[e][5]], mx_tab * 4)) ||
                (rc = buf_ensure(ctx, ctx->scratch[kSet[e][6]], mx_next * 4))) return fail(rc);
    }
    hipStream_t st2[2] = {s, s};
    if (overlap && chunk_end.size() > 1) {
        if (int rcs = ext_streams(ctx)) return fail(rcs);          // (the context's one place that makes streams: NECAT_SERIAL aliases and NECAT_STREAM_PRIO apply here too)
        st2[1] = ctx->stream_b;
    }
    tick("chunk plan + arenas");
    // what a chunk leaves on the device between its two halves
    struct Chunk { u32 pos = 0, n = 0; VoteMeta* d_meta = nullptr; ReadIdxMeta* d_rmeta = nullptr; u32* d_order = nullptr; i32 *d_nblk = nullptr, *d_nstrand = nullptr, *d_nplan = nullptr;
                   VoteArenas A; VoteCand* d_sel = nullptr; AsmPlanDev* d_plan = nullptr; u32 *d_tabs = nullptr, *d_next = nullptr; };
    Chunk chunks2[2];
    // ---- first half of a chunk: vote of both strands, the per-read cut, the reads' 10-mer tables - launched, not waited for
    auto launch_vote = [&](size_t ci, int e) -> int {
        hipStream_t sv = st2[e];
        Chunk& C = chunks2[e];
        C.pos = ci ? chunk_end[ci - 1] : 0u; C.n = chunk_end[ci] - C.pos;
        const u32 n = C.n, pos = C.pos;
        std::vector<VoteMeta> meta(n);
        std::vector<ReadIdxMeta> rmeta(n);
        u64 ht_tot = 0, pool_tot = 0, tab_tot = 0, next_tot = 0;
        for (u32 i = 0; i < n; ++i) {
            const u32 r = order[pos + i];
            for (int st = 0; st < 2; ++st) {
                const u64 H = std::max<u64>(1, hits[2 * (size_t)r + st]);
                u64 cap = 4; while (cap < 2 * H) cap <<= 1;
                meta[i].ht_off[st] = ht_tot; meta[i].ht_mask[st] = (u32)(cap - 1); ht_tot += cap;
                meta[i].pool_off[st] = pool_tot; meta[i].pool_cap[st] = (u32)H; pool_tot += H;
            }
            const u64 L = reads->h_seq_off[r + 1] - reads->h_seq_off[r];
            const u64 cap = L + L / 2 + 64;
            rmeta[i].tab_off = tab_tot; rmeta[i].next_off = next_tot; rmeta[i].cap = (u32)cap; rmeta[i]._pad = 0;
            tab_tot += 2 * cap; next_tot += L + 1;
        }
        char* mb = (char*)ctx->scratch[kSet[e][0]].p;
        auto carve = [&](size_t bytes) { char* q = mb; mb += (bytes + 63) & ~(size_t)63; return q; };
        C.d_meta = (VoteMeta*)carve(n * sizeof(VoteMeta));
        C.d_rmeta = (ReadIdxMeta*)carve(n * sizeof(ReadIdxMeta));
        C.d_order = (u32*)carve((size_t)n * 4);
        C.d_nblk = (i32*)carve((size_t)n * 8);
        C.d_nstrand = (i32*)carve((size_t)n * 8);
        C.d_nplan = (i32*)carve((size_t)n * 4);
        C.A.ht = (u64*)ctx->scratch[kSet[e][1]].p; C.A.pool = (VBlock*)ctx->scratch[kSet[e][2]].p; C.A.out = (VoteCand*)ctx->scratch[kSet[e][3]].p;
        C.d_sel = (VoteCand*)ctx->scratch[kSet[e][4]].p;
        C.d_plan = (AsmPlanDev*)((char*)ctx->scratch[kSet[e][4]].p + (size_t)n * NE * sizeof(VoteCand));
        C.d_tabs = (u32*)ctx->scratch[kSet[e][5]].p; C.d_next = (u32*)ctx->scratch[kSet[e][6]].p;
        if (hipMemcpyAsync(C.d_meta, meta.data(), n * sizeof(VoteMeta), hipMemcpyHostToDevice, sv) != hipSuccess ||
            hipMemcpyAsync(C.d_rmeta, rmeta.data(), n * sizeof(ReadIdxMeta), hipMemcpyHostToDevice, sv) != hipSuccess ||
            hipMemcpyAsync(C.d_order, order.data() + pos, (size_t)n * 4, hipMemcpyHostToDevice, sv) != hipSuccess ||
            hipMemsetAsync(C.A.ht, 0xFF, ht_tot * 8, sv) != hipSuccess ||
            hipMemsetAsync(C.d_tabs, 0, tab_tot * 4, sv) != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: chunk upload");
        hipLaunchKernelGGL(k_asm_vote_collect, dim3(2 * n), dim3(64), 0, sv, dref, drd, index_view(ix), (const u64*)ix->offset_list, P, (const u32*)C.d_order, (const VoteMeta*)C.d_meta, n, C.A,
                           C.d_nblk, d_err, (const u64*)d_kst);
        hipLaunchKernelGGL(k_asm_vote_eval, dim3(2 * n), dim3(64), 0, sv, dref, drd, P, (const u32*)C.d_order, (const VoteMeta*)C.d_meta, n, C.A, (const i32*)C.d_nblk, C.d_nstrand);
        hipLaunchKernelGGL(k_asm_select, dim3(n), dim3(64), 0, sv, P, (const VoteMeta*)C.d_meta, n, C.A, (const i32*)C.d_nstrand, C.d_sel, C.d_plan, C.d_nplan);
        hipLaunchKernelGGL(k_asm_read_index, dim3(n), dim3(64), 0, sv, drd, (const u32*)C.d_order, (const ReadIdxMeta*)C.d_rmeta, n, C.d_tabs, C.d_next);
        if (hipGetLastError() != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: vote kernels launch failed");
        return NECAT_OK;
    };
    // ---- second half: the planned pairs' match counts, matches, chains; the chunk's plan to the host
    auto finish_chunk = [&](int e) -> int {
        hipStream_t sv = st2[e];
        Chunk& C = chunks2[e];
        const u32 n = C.n, pos = C.pos;
        std::vector<i32> nplan(n);
        int herr = 0;
        if (hipMemcpyAsync(nplan.data(), C.d_nplan, (size_t)n * 4, hipMemcpyDeviceToHost, sv) != hipSuccess || hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, sv) != hipSuccess ||
            hipStreamSynchronize(sv) != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: vote kernels failed: %s", hipGetErrorString(hipGetLastError()));
        if (herr) return set_err(ctx, NECAT_ERR_CAPACITY, "asm plan: vote scratch overflow (code %d)", herr);
        tick("vote + select + read index");
        if (const char* dump = getenv("NECAT_ASM_DUMP_VOTES")) {
            // tests/host_core/check_asm_plan.cpp: per read {read id, candidates of both strands, kept}, then the ranked candidates (6 ints each)
            std::vector<VoteCand> hsel((size_t)n * NE);
            std::vector<i32> hns((size_t)n * 2);
            if (hipMemcpy(hsel.data(), C.d_sel, hsel.size() * sizeof(VoteCand), hipMemcpyDeviceToHost) == hipSuccess &&
                hipMemcpy(hns.data(), C.d_nstrand, hns.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
                if (FILE* f = fopen(dump, "ab")) {
                    for (u32 i = 0; i < n; ++i) {
                        const i32 tot = hns[2 * (size_t)i] + hns[2 * (size_t)i + 1], kept = std::min<i32>(tot, NE);
                        const i32 hdr[3] = {(i32)order[pos + i], tot, kept};
                        fwrite(hdr, 4, 3, f);
                        fwrite(hsel.data() + (size_t)i * NE, sizeof(VoteCand), (size_t)kept, f);
                    }
                    fclose(f);
                }
            }
        }
        std::vector<PairMeta> pairs;
        for (u32 i = 0; i < n; ++i) for (i32 q = 0; q < nplan[i]; ++q) { PairMeta pm; pm.read_i = i; pm.slot = (u32)q; pm.seed_off = 0; pairs.push_back(pm); }
        const u32 np = (u32)pairs.size();
        tot_pairs += np;
        std::vector<AsmPlanDev> hplan;
        int rc2;
        if (np) {
            if ((rc2 = buf_ensure(ctx, ctx->scratch[SC_ASM_PAIRS], (size_t)np * (sizeof(PairMeta) + 8) + 256))) return rc2;
            PairMeta* d_pairs = (PairMeta*)ctx->scratch[SC_ASM_PAIRS].p;
            u32* d_counts = (u32*)((char*)d_pairs + (((size_t)np * sizeof(PairMeta) + 63) & ~(size_t)63));
            u32* d_nmem = d_counts + np;
            if (hipMemcpyAsync(d_pairs, pairs.data(), (size_t)np * sizeof(PairMeta), hipMemcpyHostToDevice, sv) != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: pair upload");
            hipLaunchKernelGGL(k_asm_seeds<false>, dim3(np), dim3(64), 0, sv, dref, drd, (const u32*)C.d_order, (const ReadIdxMeta*)C.d_rmeta, (const u32*)C.d_tabs, (const u32*)C.d_next, (const u8*)d_occ,
                               (const AsmPlanDev*)C.d_plan, NE, (const PairMeta*)d_pairs, np, d_counts, (AsmSeed*)nullptr, (AsmMem*)nullptr, (AsmMem*)nullptr, (u32*)nullptr);
            std::vector<u32> counts(np);
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(counts.data(), d_counts, (size_t)np * 4, hipMemcpyDeviceToHost, sv) != hipSuccess || hipStreamSynchronize(sv) != hipSuccess)
                return set_err(ctx, NECAT_ERR_DEVICE, "k_asm_seeds<count> failed: %s", hipGetErrorString(hipGetLastError()));
            tick("match counts");
            const size_t per_seed = sizeof(AsmSeed) + 2 * sizeof(AsmMem) + 16;
            {   // the arena once per chunk, for its largest batch (+ a quarter: the next chunk's is about as large)
                u64 mx = 0, so = 0;
                for (u32 b = 0; b < np; ++b) { if (so && so + counts[b] > budget_seeds) { mx = std::max(mx, so); so = 0; } so += counts[b]; }
                mx = std::max(mx, so);
                if (std::max<u64>(1, mx) * per_seed + 256 > ctx->scratch[SC_ASM_SEEDS].cap && (rc2 = buf_ensure(ctx, ctx->scratch[SC_ASM_SEEDS], (std::max<u64>(1, mx) + mx / 4) * per_seed + 256))) return rc2;
            }
            for (u32 b0 = 0; b0 < np;) {
                u64 so = 0; u32 b1 = b0;
                while (b1 < np && (b1 == b0 || so + counts[b1] <= budget_seeds)) { pairs[b1].seed_off = so; so += counts[b1]; ++b1; }
                tot_seeds += so;
                char* sb = (char*)ctx->scratch[SC_ASM_SEEDS].p;
                AsmSeed* d_seeds = (AsmSeed*)sb; sb += ((so * sizeof(AsmSeed)) + 63) & ~(size_t)63;
                AsmMem* d_mems = (AsmMem*)sb; sb += ((so * sizeof(AsmMem)) + 63) & ~(size_t)63;
                AsmMem* d_tmp = (AsmMem*)sb; sb += ((so * sizeof(AsmMem)) + 63) & ~(size_t)63;
                i32* d_chain = (i32*)sb;
                const u32 nb = b1 - b0;
                if (hipMemcpyAsync(d_pairs + b0, pairs.data() + b0, (size_t)nb * sizeof(PairMeta), hipMemcpyHostToDevice, sv) != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: pair upload");
                hipLaunchKernelGGL(k_asm_seeds<true>, dim3(nb), dim3(64), 0, sv, dref, drd, (const u32*)C.d_order, (const ReadIdxMeta*)C.d_rmeta, (const u32*)C.d_tabs, (const u32*)C.d_next, (const u8*)d_occ,
                                   (const AsmPlanDev*)C.d_plan, NE, (const PairMeta*)(d_pairs + b0), nb, (u32*)nullptr, d_seeds, d_mems, d_tmp, d_nmem + b0);
                hipLaunchKernelGGL(k_asm_chain, dim3(nb), dim3(64), 0, sv, (const PairMeta*)(d_pairs + b0), nb, (const AsmMem*)d_mems, (const u32*)(d_nmem + b0), d_chain, C.d_plan, NE);
                if (hipGetLastError() != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: range kernels launch failed");
                if (b1 < np && hipStreamSynchronize(sv) != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: range kernels failed: %s", hipGetErrorString(hipGetLastError()));
                b0 = b1;
            }
            hplan.resize((size_t)n * NE);
            if (hipMemcpyAsync(hplan.data(), C.d_plan, (size_t)n * NE * sizeof(AsmPlanDev), hipMemcpyDeviceToHost, sv) != hipSuccess || hipStreamSynchronize(sv) != hipSuccess)
                return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: range kernels failed: %s", hipGetErrorString(hipGetLastError()));
            tick("matches + chains");
        }
        for (u32 i = 0; i < n; ++i) {
            std::vector<necat_asm_plan>& dst = per_read[order[pos + i]];
            dst.resize((size_t)nplan[i]);
            for (i32 q = 0; q < nplan[i]; ++q) {
                const AsmPlanDev& en = hplan[(size_t)i * NE + (size_t)q];
                necat_asm_plan& o = dst[(size_t)q];
                o.qid = (int32_t)order[pos + i] + read_start_id; o.sid = en.sid + ref_start_id; o.sdir = en.sdir; o.qoff = en.qoff; o.soff = en.soff; o.score = en.score; o.ssize = en.ssize;
            }
            tot_plans += (u64)nplan[i];
        }
        return NECAT_OK;
    };
    {
        const size_t nch = chunk_end.size();
        const int two = (overlap && nch > 1) ? 1 : 0;
        auto drain = [&]() { (void)hipStreamSynchronize(st2[0]); (void)hipStreamSynchronize(st2[1]); };       // nothing in flight when an error returns
        if ((rc = launch_vote(0, 0))) { drain(); return fail(rc); }
        for (size_t ci = 0; ci < nch; ++ci) {
            const int e = two ? (int)(ci & 1) : 0;
            if (two && ci + 1 < nch && (rc = launch_vote(ci + 1, e ^ 1))) { drain(); return fail(rc); }
            if ((rc = finish_chunk(e))) { drain(); return fail(rc); }
            if (!two && ci + 1 < nch && (rc = launch_vote(ci + 1, 0))) { drain(); return fail(rc); }
        }
    }
    necat_asm_plan* res = (necat_asm_plan*)result_alloc(std::max<u64>(1, tot_plans) * sizeof(necat_asm_plan));
    if (!res) return fail(set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"));
    u64 at = 0;
    for (u32 r = 0; r < nreads; ++r) { fo[r] = at; for (const necat_asm_plan& e : per_read[r]) res[at++] = e; }
    fo[nreads] = at;
    if (g_trace & 2) fprintf(stderr, "[necat] asm plan: %u reads, %lu planned pairs, %lu matches, %.2f ms\n", nreads, (unsigned long)tot_pairs, (unsigned long)tot_seeds,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    *out = res; *first = fo;
    return NECAT_OK;
}
