// stage_seed.inl - seeding (seed_kernels.h): find_impl behind necat_find_candidates.
// One of the stage files of libnecat_hip.so's single translation unit: necat_hip.hip includes them in order, inside its extern "C" block, after the
// context / knob / result-pool code they all use (the kernels are header templates and the stages share host helpers: one device code object, one 30 s build).

// ------------------------------------------------------------------------------------------ seeding

namespace {
// candidates left on the device for necat_map_pair: array in ascending read order + the first candidate of
// every read that has any (the groups of the containment filter) + the total
struct DevCands { const necat_candidate* d = nullptr; uint64_t n = 0; std::vector<u64> group_off; };

void fill_groups(DevCands* dev, const std::vector<u64>& by_read, u32 nreads)
{
    dev->group_off.clear();
    for (u32 r = 0; r < nreads; ++r) if (by_read[r + 1] > by_read[r]) dev->group_off.push_back(by_read[r]);
    dev->group_off.push_back(by_read[nreads]);
    if (dev->group_off.size() == 1) dev->group_off.insert(dev->group_off.begin(), 0);
}

// the query reads one rank of a multi-GPU job processes: chunks of `chunk` reads, chunk c in slot c % nparts
// (one rank of a sharded call: slots [rank, rank + 1) of nranks; a share of a scheduled volume pair: slots [lo, hi) of `nparts`
// - interleaved either way, because a read late in a volume sees more subjects in the self pair, word_finder.c:121-127)
struct ReadSel {
    int lo = 0, hi = 1, nparts = 1, chunk = 64;
    bool always = false;            // apply the slot test even when nparts == 1 (an empty share selects nothing)
    bool has(u32 r) const
    {
        if (nparts <= 1 && !always) return true;
        const int sl = (int)((r / (u32)chunk) % (u32)nparts);
        return sl >= lo && sl < hi;
    }
};

int find_impl(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
              int read_start_id, int ref_start_id, int pairwise, const necat_map_options* opt,
              necat_candidate** out, uint64_t* n_out, DevCands* dev, const ReadSel* sel = nullptr)
{
    if (opt->kmer_size != ix->k) return set_err(ctx, NECAT_ERR_ARG, "index was built for k=%d, options say %d", ix->k, opt->kmer_size);
    if (opt->scan_window < 1 || opt->block_size < 1 || opt->block_size > 32767)
        return set_err(ctx, NECAT_ERR_ARG, "scan_window/block_size out of range (block offsets are 16-bit, word_finder_aux.h:21)");
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const u32 nreads = (u32)reads->nseq;
    if (nreads == 0) return NECAT_OK;
    ArenaUse in_use(ctx, {SC_SEED_POOL, SC_SEED_CHAIN, SC_SEED_OUT, SC_SEED_HT, SC_SEED_META});      // (buf_ensure_lend: held for the length of this call)
    DevVolume dref = dev_view(ref), drd = dev_view(reads);
    NECAT_HIP(ctx, hipEventRecord(ctx->ev[0], s));
    int rc;
    auto t_prev = std::chrono::steady_clock::now();
    auto tick = [&](const char* what) {
        if (!(g_trace & 2)) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[necat] seeding %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // ---- pass 1: hit counts per read-strand
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_MISC], (size_t)nreads * 8 + 128))) return rc;
    u32* d_hits = (u32*)ctx->scratch[SC_MISC].p;
    int* d_err = (int*)((char*)ctx->scratch[SC_MISC].p + (((size_t)nreads * 8 + 63) & ~(size_t)63));   // error flag of the seeding kernels
    // the table words k_seed_hits fetches are kept for the collection pass (seed_kst_base): one lookup per sampled k-mer, not two
    u64* d_kst = nullptr;
    if (g_seed_wave && g_seed_kst) {
        if ((rc = buf_ensure(ctx, ctx->scratch[SC_SEED_KST], 2 * (reads->nbases / (u64)opt->scan_window + nreads + 2) * 8))) return rc;
        d_kst = (u64*)ctx->scratch[SC_SEED_KST].p;
    }
    hipLaunchKernelGGL(k_seed_hits, dim3(grid_for((u64)nreads * 64, 256)), dim3(256), 0, s, drd, index_view(ix),
                       opt->kmer_size, opt->scan_window, 0u, nreads, d_hits, d_kst);
    NECAT_CHECK_LAUNCH(ctx, "k_seed_hits");
    // pinned host scratch: [hits: 2 u32 per read][order: u32 per read][SeedMeta per read] - pageable copies cost more than the plan
    {
        const size_t need = (size_t)nreads * (8 + 4 + sizeof(SeedMeta)) + 256;
        if (need > ctx->pin_plan_cap) {
            if (ctx->pin_plan) (void)hipHostFree(ctx->pin_plan);
            ctx->pin_plan = nullptr; ctx->pin_plan_cap = 0;
            if (hipHostMalloc(&ctx->pin_plan, need + need / 4, hipHostMallocDefault) != hipSuccess) return set_err(ctx, NECAT_ERR_MEMORY, "pinned host scratch (%zu bytes)", need);
            ctx->pin_plan_cap = need + need / 4;
        }
    }
    u32* hits = (u32*)ctx->pin_plan;
    u32* order = hits + (size_t)nreads * 2;
    SeedMeta* meta_all = (SeedMeta*)(((uintptr_t)(order + nreads) + 63) & ~(uintptr_t)63);
    NECAT_HIP(ctx, hipMemcpyAsync(hits, d_hits, (size_t)nreads * 8, hipMemcpyDeviceToHost, s));
    NECAT_HIP(ctx, hipStreamSynchronize(s));
    tick("hits kernel + copy");
    // ---- plan: reads in descending work order, chunks bounded by a scratch budget
    u32 nsel = 0;
    {
        // descending work, ascending read id inside equal work (only this rank's reads): a stable LSD radix sort of the
        // complemented hit counts, 3 x 11 bits (std::sort of the same keys took ~1 ms for 23 k reads)
        std::vector<u32> key(nreads), ida(nreads), idb(nreads);
        for (u32 r = 0; r < nreads; ++r)
            if (!sel || sel->has(r)) { key[r] = 0xffffffffu - std::max(hits[2 * (size_t)r], hits[2 * (size_t)r + 1]); ida[nsel++] = r; }
        u32* src = ida.data(); u32* dst = idb.data();
        for (int pass = 0; pass < 3; ++pass) {
            const int sh = 11 * pass;
            u32 cnt[2049] = {0};
            for (u32 i = 0; i < nsel; ++i) ++cnt[((key[src[i]] >> sh) & 2047u) + 1];
            for (int b = 0; b < 2048; ++b) cnt[b + 1] += cnt[b];
            for (u32 i = 0; i < nsel; ++i) dst[cnt[(key[src[i]] >> sh) & 2047u]++] = src[i];
            std::swap(src, dst);
        }
        for (u32 i = 0; i < nsel; ++i) order[i] = src[i];
    }
    ctx->shard_tm.reads_local = nsel;
    {   // the terms of SURVEY 8d's B_seed for this call (bench.py: roofline_seed)
        u64 lk = 0, ht = 0, bs = 0;
        const bool have_off = reads->h_seq_off.size() == (size_t)nreads + 1;
        for (u32 i = 0; i < nsel; ++i) {
            const u32 r = order[i];
            ht += (u64)hits[2 * (size_t)r] + hits[2 * (size_t)r + 1];
            if (have_off) { const u64 L = reads->h_seq_off[r + 1] - reads->h_seq_off[r]; bs += L; if (L >= (u64)opt->kmer_size) lk += (L - (u64)opt->kmer_size) / (u64)opt->scan_window + 1; }
        }
        ctx->tm.seed_bases = 2 * bs; ctx->tm.seed_lookups = 2 * lk; ctx->tm.seed_hits = ht; ctx->tm.seed_cands = 0;
    }
    if (nsel == 0) {
        if (dev) { dev->n = 0; dev->d = nullptr; dev->group_off.assign(2, 0); }
        else { *out = (necat_candidate*)result_alloc(sizeof(necat_candidate)); *n_out = 0; }
        ctx->tm.seed_ms = 0;
        return NECAT_OK;
    }
    const u64 budget_hits = g_seed_budget;   // default ~48 M pool blocks (~13 GB of SBlocks) per chunk
    SeedParams P;
    P.k = opt->kmer_size; P.z = opt->scan_window; P.block_size = opt->block_size; P.s_cutoff = opt->block_score_cutoff;
    P.align_cutoff = opt->align_size_cutoff; P.num_candidates = opt->num_candidates; P.job = opt->job; P.pairwise = pairwise;
    P.read_start_id = read_start_id; P.ref_start_id = ref_start_id;
    P.debug_phase = getenv("NECAT_SEED_DEBUG") ? atoi(getenv("NECAT_SEED_DEBUG")) : 0;
    P.chain_wave = getenv("NECAT_CHAIN_WAVE") ? atoi(getenv("NECAT_CHAIN_WAVE")) : 1;
    NECAT_HIP(ctx, hipMemsetAsync(d_err, 0, 4, s));
    u32 pos = 0;
    std::vector<i32> ncands_by_order(nsel, 0);
    // every chunk's compacted candidates stay on the device (SC_SEED_ALL), in ORDER-index space
    u64 packed_total = 0;
    std::vector<u64> packed_off(nsel + 1, 0);
    while (pos < nsel) {
        u64 acc = 0; u32 hi = pos;
        auto both = [&](u32 r) { return (u64)hits[2 * (size_t)r] + hits[2 * (size_t)r + 1] + 2; };
        while (hi < nsel && (hi == pos || acc + both(order[hi]) <= budget_hits)) { acc += both(order[hi]); ++hi; }
        const u32 n = hi - pos;
        SeedMeta* meta = meta_all + pos;
        u64 ht_tot = 0, pool_tot = 0, chain_tot = 0, out_tot = 0;
        for (u32 i = 0; i < n; ++i) {
            const u32 r = order[pos + i];
            SeedMeta& m = meta[i];
            for (int st = 0; st < 2; ++st) {
                const u64 H = std::max<u64>(1, hits[2 * (size_t)r + st]);
                u64 cap = 4; while (cap < 2 * H) cap <<= 1;
                m.ht_off[st] = ht_tot; m.ht_mask[st] = (u32)(cap - 1); ht_tot += cap;
                m.pool_off[st] = pool_tot; m.pool_cap[st] = (u32)H; pool_tot += H;
                // chain scratch per strand: the two strands of a read are evaluated by two waves at the same time
                m.chain_off[st] = chain_tot; m.cs_cap[st] = (u32)(H + 1); chain_tot += H + 1;
            }
            const u64 oc = (u64)hits[2 * (size_t)r] + hits[2 * (size_t)r + 1] + 2;
            m.out_off = out_tot; m.out_cap = (u32)oc; m.out_cap0 = hits[2 * (size_t)r] + 1; out_tot += oc;
        }
        if ((rc = buf_ensure(ctx, ctx->scratch[SC_SEED_META], n * sizeof(SeedMeta) + (size_t)n * (8 + 4 + 4 + 8 + 8) + 64)) ||
            (rc = buf_ensure(ctx, ctx->scratch[SC_SEED_HT], ht_tot * 8)) ||
            // (the block pool and the chain scratch take over the index build's split buffers, idle until the next build: runtime.h)
            (rc = buf_ensure_lend(ctx, SC_SEED_POOL, pool_tot * sizeof(SBlock), {SC_PART2, SC_TMPLIST})) ||
            (rc = buf_ensure_lend(ctx, SC_SEED_CHAIN, chain_tot * (8 + 16 + 8 + sizeof(DevCand)), {SC_TMPLIST, SC_PART2})) ||
            (rc = buf_ensure_lend(ctx, SC_SEED_OUT, out_tot * sizeof(DevCand), {SC_TMPLIST, SC_PART2}))) { return rc; }
        tick("plan + buffers");
        char* mb = (char*)ctx->scratch[SC_SEED_META].p;
        SeedMeta* d_meta = (SeedMeta*)mb; mb += n * sizeof(SeedMeta);
        u64* d_final = (u64*)mb; mb += (size_t)n * 8;
        i32* d_nblk = (i32*)mb; mb += (size_t)n * 8;
        u32* d_order = (u32*)mb; mb += (size_t)n * 4;
        i32* d_ncand = (i32*)mb; mb += (size_t)n * 4;
        i32* d_nstrand = (i32*)mb;
        SeedArenas A;
        A.ht = (u64*)ctx->scratch[SC_SEED_HT].p;
        A.pool = (SBlock*)ctx->scratch[SC_SEED_POOL].p;
        char* cb = (char*)ctx->scratch[SC_SEED_CHAIN].p;
        A.cs = (u64*)cb; cb += chain_tot * 8;
        A.u = (u64*)cb; cb += chain_tot * 8;
        A.lcan = (DevCand*)cb; cb += chain_tot * sizeof(DevCand);
        A.f = (i32*)cb; cb += chain_tot * 4; A.p = (i32*)cb; cb += chain_tot * 4; A.t = (i32*)cb; cb += chain_tot * 4; A.v = (i32*)cb;
        A.out = (DevCand*)ctx->scratch[SC_SEED_OUT].p;
        NECAT_HIP(ctx, hipMemcpyAsync(d_meta, meta, n * sizeof(SeedMeta), hipMemcpyHostToDevice, s));
        NECAT_HIP(ctx, hipMemcpyAsync(d_order, order + pos, (size_t)n * 4, hipMemcpyHostToDevice, s));
        // the hash arena is all-empty between calls (k_seed_clear below): filled only when it is new or a failed call left it dirty
        // (only the stretch this chunk uses beyond what is known clean: a fresh 13 GB arena is not filled for a 0.3 GB chunk)
        if (ctx->seed_ht_ptr != ctx->scratch[SC_SEED_HT].p || ctx->seed_ht_cap != ctx->scratch[SC_SEED_HT].cap) {      // a new allocation (also one at the old address)
            ctx->seed_ht_ptr = ctx->scratch[SC_SEED_HT].p; ctx->seed_ht_cap = ctx->scratch[SC_SEED_HT].cap; ctx->seed_ht_clean = 0;
        }
        const size_t ht_clean_before = ctx->seed_ht_clean;
        if (ht_clean_before < ht_tot * 8) NECAT_HIP(ctx, hipMemsetAsync((char*)A.ht + ht_clean_before, 0xFF, ht_tot * 8 - ht_clean_before, s));
        const size_t ht_clean_after = std::max<size_t>(ht_clean_before, ht_tot * 8);
        ctx->seed_ht_clean = 0;      // in use: clean again once this chunk's kernels (k_seed_clear last) are known to have run
        if (g_seed_wave)
            hipLaunchKernelGGL(k_seed_collect_wave, dim3(2 * n), dim3(64), 0, s, dref, drd, index_view(ix), (const u64*)ix->offset_list,
                               P, (const u32*)d_order, (const SeedMeta*)d_meta, n, A, d_nblk, d_err, (const u64*)d_kst);
        else
#if NECAT_XCHECK
            hipLaunchKernelGGL(k_seed_collect, dim3(grid_for((u64)2 * n, 64)), dim3(64), 0, s, dref, drd, index_view(ix), (const u64*)ix->offset_list,
                               P, (const u32*)d_order, (const SeedMeta*)d_meta, n, A, d_nblk, d_err);
#else
            NECAT_RETIRED(ctx, "the lane-per-strand seed collection (NECAT_SEED_WAVE=0)");
#endif
        NECAT_CHECK_LAUNCH(ctx, "k_seed_collect");
        static const bool fused_clear = !getenv("NECAT_SEED_CLEAR_KERNEL");        // (A/B: the slots cleared by a launch of their own, as in round 3)
        hipLaunchKernelGGL(k_seed_eval, dim3(2 * n), dim3(64), 0, s, dref, drd, P, (const u32*)d_order, (const SeedMeta*)d_meta, n, A,
                           (const i32*)d_nblk, d_nstrand, d_err, fused_clear && P.debug_phase != 1 ? 1 : 0);
        NECAT_CHECK_LAUNCH(ctx, "k_seed_eval");
        if (!fused_clear || P.debug_phase == 1) {
            hipLaunchKernelGGL(k_seed_clear, dim3(2 * n), dim3(64), 0, s, (const SeedMeta*)d_meta, n, A, (const i32*)d_nblk);
            NECAT_CHECK_LAUNCH(ctx, "k_seed_clear");
        }
        hipLaunchKernelGGL(k_seed_finish, dim3(grid_for(n, 64)), dim3(64), 0, s, P, (const SeedMeta*)d_meta, n, A, (const i32*)d_nstrand, d_ncand);
        NECAT_CHECK_LAUNCH(ctx, "k_seed_finish");
        std::vector<i32> nc(n);
        NECAT_HIP(ctx, hipMemcpyAsync(nc.data(), d_ncand, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        int herr = 0;
        NECAT_HIP(ctx, hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, s));
        NECAT_HIP(ctx, hipStreamSynchronize(s));
        if (herr) { return set_err(ctx, NECAT_ERR_CAPACITY, "seeding scratch overflow (code %d)", herr); }
        ctx->seed_ht_clean = ht_clean_after;
#ifdef NECAT_SEED_PROF
        {   // tools/seed_prof.sh: cycles of lane 0 per phase of k_seed_eval, summed over the waves
            unsigned long long h[32], z[32] = {0};
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_seed_prof), sizeof h) == hipSuccess) {
                static const char* nm[10] = {"block test", "A seed lists", "B vote", "C anchor", "D gather", "E sort", "emit (lane 0)", "chain DP (wave)", "evaluations", "exit"};
                unsigned long long tot = 0; for (int q = 0; q < 10; ++q) if (q != 8) tot += h[q];
                for (int q = 0; q < 10; ++q) fprintf(stderr, "[seed prof] %-16s %14llu %5.1f %%\n", nm[q], h[q], q == 8 ? 0.0 : 100.0 * h[q] / (double)tot);
                fprintf(stderr, "[seed prof] longest wave %llu cycles, most evaluations in a wave %llu\n", h[10], h[11]);
                if (h[26]) for (int q = 0; q < 10; ++q) fprintf(stderr, "[seed prof] waves over 3 M cycles (%llu): %-16s %12llu per wave\n", h[26], nm[q], h[16 + q] / h[26]);
            }
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_seed_prof), z, sizeof z);
        }
#endif
        tick("collect + eval kernels");
        if (pos == 0 && hi == nsel) {
            // the usual case, one chunk: pack on the device straight into ascending read order and copy
            // into the (pinned) result block
            std::vector<u64> by_read((size_t)nreads + 1, 0), foff(n);
            for (u32 i = 0; i < n; ++i) by_read[order[i] + 1] = (u64)nc[i];
            for (u32 r = 0; r < nreads; ++r) by_read[r + 1] += by_read[r];
            for (u32 i = 0; i < n; ++i) foff[i] = by_read[order[i]];
            const u64 tot = by_read[nreads];
            necat_candidate* res = dev ? nullptr : (necat_candidate*)result_alloc(std::max<u64>(1, tot) * sizeof(necat_candidate));
            if (!dev && !res) { return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"); }
            if (dev) { dev->n = tot; fill_groups(dev, by_read, nreads); }
            if (tot) {
                if ((rc = buf_ensure(ctx, ctx->scratch[SC_SEED_FINAL], tot * sizeof(necat_candidate)))) { necat_free(res); return rc; }
                necat_candidate* d_dst = (necat_candidate*)ctx->scratch[SC_SEED_FINAL].p;
                hipError_t e1 = hipMemcpyAsync(d_final, foff.data(), (size_t)n * 8, hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(k_pack_cands, dim3(grid_for((u64)n * 64, 256)), dim3(256), 0, s, (const DevCand*)A.out, (const SeedMeta*)d_meta,
                                   (const i32*)d_ncand, (const u64*)d_final, n, read_start_id, ref_start_id, d_dst);
                hipError_t e2 = hipGetLastError();
                if (dev) dev->d = d_dst;
                hipError_t e3 = dev ? hipSuccess : hipMemcpyAsync(res, d_dst, tot * sizeof(necat_candidate), hipMemcpyDeviceToHost, s);
                hipError_t e4 = hipEventRecord(ctx->ev[1], s);
                hipError_t e5 = hipStreamSynchronize(s);
                for (hipError_t e : {e1, e2, e3, e4, e5})
                    if (e != hipSuccess) { necat_free(res); return set_err(ctx, NECAT_ERR_DEVICE, "seeding result copy: %s", hipGetErrorString(e)); }
            } else { NECAT_HIP(ctx, hipEventRecord(ctx->ev[1], s)); NECAT_HIP(ctx, hipStreamSynchronize(s)); }
            ctx->tm.seed_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
            ctx->tm.seed_cands = tot;
            tick("pack + copy to host");
            if (!dev) { *out = res; *n_out = tot; }
            return NECAT_OK;
        }
        // several chunks: pack this chunk's candidates behind the earlier ones, on the device
        std::vector<u64> foff(n + 1, 0);
        for (u32 i = 0; i < n; ++i) foff[i + 1] = foff[i] + (u64)nc[i];
        const u64 tot = foff[n];
        if (tot) {
            if ((rc = buf_grow(ctx, ctx->scratch[SC_SEED_ALL], (packed_total + tot) * sizeof(necat_candidate), packed_total * sizeof(necat_candidate), s))) { return rc; }
            necat_candidate* d_dst = (necat_candidate*)ctx->scratch[SC_SEED_ALL].p + packed_total;
            NECAT_HIP(ctx, hipMemcpyAsync(d_final, foff.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_pack_cands, dim3(grid_for((u64)n * 64, 256)), dim3(256), 0, s, (const DevCand*)A.out, (const SeedMeta*)d_meta,
                               (const i32*)d_ncand, (const u64*)d_final, n, read_start_id, ref_start_id, d_dst);
            NECAT_CHECK_LAUNCH(ctx, "k_pack_cands");
            NECAT_HIP(ctx, hipStreamSynchronize(s));       // foff / nc are host vectors of this iteration
        }
        for (u32 i = 0; i < n; ++i) { ncands_by_order[pos + i] = nc[i]; packed_off[pos + i] = packed_total + foff[i]; }
        packed_total += tot;
        tick("pack");
        pos = hi;
    }
    // ---- ascending read id: one move on the device, one copy into the (pinned) result block
    const u64 total = packed_total;
    necat_candidate* res = dev ? nullptr : (necat_candidate*)result_alloc(std::max<u64>(1, total) * sizeof(necat_candidate));
    if (!dev && !res) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
    std::vector<u64> by_read((size_t)nreads + 1, 0), dst_off(nsel);
    for (u32 i = 0; i < nsel; ++i) by_read[order[i] + 1] = (u64)ncands_by_order[i];
    for (u32 r = 0; r < nreads; ++r) by_read[r + 1] += by_read[r];
    for (u32 i = 0; i < nsel; ++i) dst_off[i] = by_read[order[i]];
    if (dev) { dev->n = total; fill_groups(dev, by_read, nreads); }
    if (total) {
        int rc2;
        if ((rc2 = buf_ensure(ctx, ctx->scratch[SC_SEED_FINAL], total * sizeof(necat_candidate))) ||
            (rc2 = buf_ensure(ctx, ctx->scratch[SC_SEED_META], (size_t)nreads * 20 + 64))) { necat_free(res); return rc2; }
        char* mb = (char*)ctx->scratch[SC_SEED_META].p;
        u64* d_src = (u64*)mb; mb += (size_t)nreads * 8;
        u64* d_dsto = (u64*)mb; mb += (size_t)nreads * 8;
        i32* d_cnt = (i32*)mb;
        necat_candidate* d_fin = (necat_candidate*)ctx->scratch[SC_SEED_FINAL].p;
        hipError_t e[7];
        e[0] = hipMemcpyAsync(d_src, packed_off.data(), (size_t)nsel * 8, hipMemcpyHostToDevice, s);
        e[1] = hipMemcpyAsync(d_dsto, dst_off.data(), (size_t)nsel * 8, hipMemcpyHostToDevice, s);
        e[2] = hipMemcpyAsync(d_cnt, ncands_by_order.data(), (size_t)nsel * 4, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(k_move_cands, dim3(grid_for((u64)nsel * 64, 256)), dim3(256), 0, s, (const necat_candidate*)ctx->scratch[SC_SEED_ALL].p,
                           (const u64*)d_src, (const u64*)d_dsto, (const i32*)d_cnt, nsel, d_fin);
        e[3] = hipGetLastError();
        if (dev) dev->d = d_fin;
        e[4] = dev ? hipSuccess : hipMemcpyAsync(res, d_fin, total * sizeof(necat_candidate), hipMemcpyDeviceToHost, s);
        e[5] = hipEventRecord(ctx->ev[1], s);
        e[6] = hipStreamSynchronize(s);
        for (hipError_t x : e) if (x != hipSuccess) { necat_free(res); return set_err(ctx, NECAT_ERR_DEVICE, "seeding result assembly: %s", hipGetErrorString(x)); }
    } else { NECAT_HIP(ctx, hipEventRecord(ctx->ev[1], s)); NECAT_HIP(ctx, hipStreamSynchronize(s)); }
    ctx->tm.seed_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
    ctx->tm.seed_cands = total;
    tick("assemble in read order");
    if (!dev) { *out = res; *n_out = total; }
    return NECAT_OK;
}
}  // namespace

int necat_find_candidates(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                          int read_start_id, int ref_start_id, int pairwise, const necat_map_options* opt,
                          necat_candidate** out, uint64_t* n_out)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ix || !ref || !reads || !opt || !out || !n_out) return NECAT_ERR_ARG;
    *out = nullptr; *n_out = 0;
    return find_impl(ctx, ix, ref, reads, read_start_id, ref_start_id, pairwise, opt, out, n_out, nullptr);
}
