// stage_index.inl - the index build (index_kernels.h): one rank or hash-range slices, the build plan, downloads.
// One of the stage files of libnecat_hip.so's single translation unit: necat_hip.hip includes them in order, inside its extern "C" block, after the
// context / knob / result-pool code they all use (the kernels are header templates and the stages share host helpers: one device code object, one 30 s build).

// ------------------------------------------------------------------------------------------ index

namespace {
int index_build_impl(necat_ctx* ctx, necat_comm* comm, const necat_volume* ref, int kmer_size, int max_occ, necat_index** out);
}
int necat_index_build(necat_ctx* ctx, const necat_volume* ref, int kmer_size, int max_occ, necat_index** out)
{
    KnobScope knob_scope_(ctx);
    return index_build_impl(ctx, nullptr, ref, kmer_size, max_occ, out);
}

int necat_index_plan(uint64_t nbases, int kmer_size, int nranks, double link_gbs, necat_index_plan_t* out)
{
    if (!out || kmer_size < 1 || kmer_size > 15 || nranks < 1) return NECAT_ERR_ARG;
    if (link_gbs <= 0) { const char* e = getenv("NECAT_XGMI_GBS"); link_gbs = e && atof(e) > 0 ? atof(e) : 100.0; }
    const double N = (double)nbases, T = (double)(1ULL << (2 * kmer_size));
    const double scan_ms = 5.98e-9 * N, work_ms = 22.3e-9 * N;                    // (1.1 + 4.1 ms at 184 Mbp; 58 ms at 2.0 Gbp: profiles/r04_kernel_stats.md, r05_config4_human_subset.json)
    const double distinct = T * (1.0 - exp(-N / T));                              // non-zero table entries of N uniformly drawn k-mers (an upper bound for real reads)
    const double bytes = T / 64.0 * 16.0 + 8.0 * distinct + 8.0 * N;
    out->_pad = 0;
    out->replicate_ms = scan_ms + work_ms;
    out->exchange_bytes = (uint64_t)bytes;
    out->exchange_ms = nranks > 1 ? 3 * 0.05 + bytes / nranks / (link_gbs * 1e6) : 0.0;
    out->shard_ms = scan_ms + work_ms / nranks + out->exchange_ms;
    out->shard = nranks > 1 && out->shard_ms < out->replicate_ms;
    if (const char* e = getenv("NECAT_INDEX_SHARD")) out->shard = nranks > 1 && atoi(e) != 0;
    return NECAT_OK;
}

int necat_index_build_sharded(necat_ctx* ctx, necat_comm* comm, const necat_volume* ref, int kmer_size, int max_occ, necat_index** out)
{
    KnobScope knob_scope_(ctx);
    if (!comm) return NECAT_ERR_ARG;
    return index_build_impl(ctx, comm, ref, kmer_size, max_occ, out);
}

namespace {
IndexView index_view(const necat_index* ix)
{
    IndexView v; v.dense = ix->kmer_stats; v.words = (const IdxWord*)ix->words; v.compact = ix->compact;
    return v;
}

// the table's allocation: the cached one of an earlier index of this context if it is big enough (a fresh hipMalloc of
// gigabytes costs tens of ms)
int table_alloc(necat_ctx* ctx, necat_index* ix, size_t bytes)
{
    if (ctx->idx_cache[0].p && ctx->idx_cache[0].cap >= bytes) { ix->table = ctx->idx_cache[0].p; ix->stats_cap = ctx->idx_cache[0].cap; ctx->idx_cache[0] = DevBuf(); }
    else { NECAT_HIP(ctx, hipMalloc(&ix->table, bytes)); ix->stats_cap = bytes; }
    return NECAT_OK;
}

// comm != nullptr: this rank builds the slice of the table its hash range covers, then the slices are all-gathered.
// `ix` belongs to the caller (index_build_impl), which frees it with everything it holds when a step fails.
int index_build_body(necat_ctx* ctx, necat_comm* comm, const necat_volume* ref, int kmer_size, int max_occ, necat_index* ix)
{
    const double w0 = wall_ms();
    ArenaUse in_use(ctx, {SC_PART, SC_PART2, SC_SPLIT, SC_SPLIT2, SC_TMPLIST, SC_SMALL});      // (buf_ensure_lend: nobody borrows these while this build holds pointers into them)
    if (kmer_size < 1 || kmer_size > 15) return set_err(ctx, NECAT_ERR_ARG, "kmer_size %d outside 1..15 (HashBits = 30, lookup_table.h:13)", kmer_size);
    if (max_occ < 0) return set_err(ctx, NECAT_ERR_ARG, "negative kmer_cnt_cutoff");
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint64_t T = 1ULL << (2 * kmer_size);
    const uint64_t ntiles = (T + kScanTile - 1) / kScanTile;
    DevVolume vol = dev_view(ref);
    ix->k = kmer_size; ix->table_entries = T;
    int rc;
    // partition parameters: buckets of <= 2^18 table entries (1 MB of counters), at most 4096 buckets
    int PB = 2 * kmer_size - 18; if (PB > 12) PB = 12;
    const bool partitioned = PB >= 4 && ref->nbases > 0;
    const bool lds_slices = partitioned && g_index_lds;
    const u32 NB = partitioned ? (1u << PB) : 0u;
    const int pshift = 2 * kmer_size - PB;
    // hash-range sharding: rank g owns buckets [g NB / G, (g + 1) NB / G) = table entries [that << pshift); small tables
    // (k < 11) and the global-atomic fallback are built whole on every rank
    const int G = comm ? comm->nranks : 1, rk = comm ? comm->rank : 0;
    // slices + all-gather only where that is the cheaper plan (necat_index_plan): one rank builds an E. coli-size table in 5 ms, the all-gather of its
    // 3.3 GB takes longer than that for every N <= 4 - there every rank builds the whole table and nothing is exchanged
    necat_index_plan_t plan; plan.shard = 0; plan.replicate_ms = plan.shard_ms = 0;
    if (G > 1) (void)necat_index_plan(ref->nbases, kmer_size, G, 0.0, &plan);
    const bool sharded = G > 1 && lds_slices && NB >= (u32)G && plan.shard;
    ctx->shard_tm.index_sharded = sharded ? 1 : 0; ctx->shard_tm.index_plan_replicate_ms = plan.replicate_ms; ctx->shard_tm.index_plan_shard_ms = plan.shard_ms;
    if (G > 1) {
        // The plan is every rank's own arithmetic on its own environment (NECAT_INDEX_SHARD, NECAT_XGMI_GBS, NECAT_INDEX_LDS): ranks that decide
        // differently would take different collective paths and wait for each other forever - they compare their decisions first and fail together
        int mine_plan = sharded ? 1 : 0;
        std::vector<int> all(G, 0);
        if (const int rg = comm->gather(comm->user, &mine_plan, all.data(), sizeof(int))) return set_err(ctx, NECAT_ERR_COMM, "host all-gather callback failed (%d)", rg);
        for (int g = 0; g < G; ++g)
            if (all[g] != all[0])
                return set_err(ctx, NECAT_ERR_COMM, "ranks disagree on the index build plan (rank 0: %s, rank %d: %s): NECAT_INDEX_SHARD / NECAT_XGMI_GBS / NECAT_INDEX_LDS must be the same on every rank",
                               all[0] ? "slices" : "replicate", g, all[g] ? "slices" : "replicate");
    }
    const u32 b_lo = sharded ? (u32)((u64)rk * NB / G) : 0u, b_hi = sharded ? (u32)((u64)(rk + 1) * NB / G) : NB;
    ctx->shard_tm.index_local_ms = 0; ctx->shard_tm.index_exchange_ms = 0; ctx->shard_tm.index_exchange_bytes = 0;
    // A sharded build is a sequence of collective steps.  Whatever fails on ONE rank between two of them (an allocation, a launch)
    // is reported to all ranks at the next step (comm::agree) instead of leaving the peers waiting in an exchange this rank never
    // joins: the rank-local work runs in lambdas (`local_phase`, `emit_phase`) whose status is agreed on before the data moves.
    u32* cnt32 = nullptr; u64* partial = nullptr;
    u32* d_bcnt = nullptr; u64* d_bstart = nullptr; u64* d_bcur = nullptr; u64* d_part = nullptr;
    u32 bchunks = 1;
    u64 *d_part2 = nullptr, *d_sub = nullptr, *d_bbase = nullptr, *d_cbase = nullptr;
    u32 *d_kept = nullptr, *d_pres = nullptr, *d_bpres = nullptr;
    unsigned nsl = 0; u32 s0 = 0;
    unsigned long long mine[2] = {0, 0};                        // offset-list entries, non-zero table entries of this rank
    const uint64_t nchunks = (ref->nbases + kPosPerThread - 1) / kPosPerThread;
    const unsigned pass_grid = grid_for(nchunks, 256, 1u << 16);
    auto local_phase = [&]() -> int {
    if (!lds_slices) {
        if ((rc = buf_ensure(ctx, ctx->scratch[SC_CNT32], T * 4)) || (rc = buf_ensure(ctx, ctx->scratch[SC_PARTIAL], (ntiles + 1) * 8))) return rc;
        cnt32 = (u32*)ctx->scratch[SC_CNT32].p;
        partial = (u64*)ctx->scratch[SC_PARTIAL].p;
    }
    if (!lds_slices) {      // the dense reference layout (small tables, NECAT_INDEX_LDS=0); the slice build sizes its sparse table later
        if ((rc = table_alloc(ctx, ix, T * 8))) return rc;
        ix->kmer_stats = (uint64_t*)ix->table;
    }
    NECAT_HIP(ctx, hipEventRecord(ctx->ev[0], s));
    if (!lds_slices) NECAT_HIP(ctx, hipMemsetAsync(cnt32, 0, T * 4, s));
    if (partitioned) {
        // SC_PART ends up as the index's offset list (emit_phase) and comes back through idx_cache[1] when that index is released
        if (lds_slices && !sharded && ctx->scratch[SC_PART].cap < (ref->nbases + 1) * 8 && ctx->idx_cache[1].cap >= (ref->nbases + 1) * 8) {
            if (ctx->scratch[SC_PART].p) (void)hipFree(ctx->scratch[SC_PART].p);
            ctx->scratch[SC_PART] = ctx->idx_cache[1]; ctx->idx_cache[1] = DevBuf();
        }
        if ((rc = buf_ensure(ctx, ctx->scratch[SC_SMALL], (size_t)NB * 4 + (size_t)(NB + 1) * 8 * 2 + 64)) ||
            (rc = buf_ensure(ctx, ctx->scratch[SC_PART], (ref->nbases + 1) * 8))) return rc;
        char* sb = (char*)ctx->scratch[SC_SMALL].p;
        d_bstart = (u64*)sb; sb += (size_t)(NB + 1) * 8; d_bcur = (u64*)sb; sb += (size_t)(NB + 1) * 8; d_bcnt = (u32*)sb;
        d_part = (u64*)ctx->scratch[SC_PART].p;
        const unsigned pgrid = (unsigned)((ref->nbases + kPartPosPerBlock - 1) / kPartPosPerBlock);
        NECAT_HIP(ctx, hipMemsetAsync(d_bcnt, 0, (size_t)NB * 4, s));
        // the PB partition bits in two splits of <= 6 bits (index_kernels.h): volume -> coarse buckets (in SC_PART2), coarse ->
        // fine buckets (in SC_PART); at most 64 buckets: one split
        const int bits2 = PB > 6 ? PB - 6 : 0;
        const u32 NC = NB >> bits2;
        if ((rc = buf_ensure(ctx, ctx->scratch[SC_SPLIT], (size_t)(NC + 1) * 8 * kCurStride + (size_t)(NC + 1) * 4 + 64)) ||
            (bits2 && (rc = buf_ensure_lend(ctx, SC_PART2, (ref->nbases + 1) * 8 + ((u64)NB * kSubs + 1) * 16 + (u64)NB * kSubs * 4 + 256, {SC_SEED_POOL, SC_SEED_CHAIN, SC_SEED_OUT})))) return rc;
        u64* d_ccur = (u64*)ctx->scratch[SC_SPLIT].p;
        u32* d_tpre = (u32*)(d_ccur + (size_t)(NC + 1) * kCurStride);
        hipLaunchKernelGGL(k_part_hist, dim3(pgrid), dim3(kPartThreads), NB * 2, s, vol, kmer_size, pshift, NB, b_lo, b_hi, d_bcnt);
        NECAT_CHECK_LAUNCH(ctx, "k_part_hist");
        hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, s, (const u32*)d_bcnt, NB, d_bstart, d_bcur, bits2, d_ccur, d_tpre);
        NECAT_CHECK_LAUNCH(ctx, "k_bucket_scan");
        const unsigned tgrid = (unsigned)((ref->nbases + kSplitTile - 1) / kSplitTile);
        if (bits2) {
            u64* d_coarse = (u64*)ctx->scratch[SC_PART2].p;
            if (g_split_threads == 512) hipLaunchKernelGGL(k_split_bases<512>, dim3(tgrid), dim3(512), 0, s, vol, kmer_size, pshift, b_lo, b_hi, bits2, d_ccur, kCurStride, d_coarse);
            else hipLaunchKernelGGL(k_split_bases<256>, dim3(tgrid), dim3(256), 0, s, vol, kmer_size, pshift, b_lo, b_hi, bits2, d_ccur, kCurStride, d_coarse);
            NECAT_CHECK_LAUNCH(ctx, "k_split_bases");
            if (g_split_threads == 512) hipLaunchKernelGGL(k_split_recs<512>, dim3(tgrid + NC), dim3(512), 0, s, (const u64*)d_coarse, (const u64*)d_bstart, (const u32*)d_tpre, (int)NC, pshift, bits2, d_bcur, d_part);
            else hipLaunchKernelGGL(k_split_recs<256>, dim3(tgrid + NC), dim3(256), 0, s, (const u64*)d_coarse, (const u64*)d_bstart, (const u32*)d_tpre, (int)NC, pshift, bits2, d_bcur, d_part);
            NECAT_CHECK_LAUNCH(ctx, "k_split_recs");
        } else {
            if (g_split_threads == 512) hipLaunchKernelGGL(k_split_bases<512>, dim3(tgrid), dim3(512), 0, s, vol, kmer_size, pshift, b_lo, b_hi, 0, d_bcur, 1, d_part);
            else hipLaunchKernelGGL(k_split_bases<256>, dim3(tgrid), dim3(256), 0, s, vol, kmer_size, pshift, b_lo, b_hi, 0, d_bcur, 1, d_part);
            NECAT_CHECK_LAUNCH(ctx, "k_split_bases");
        }
    }
    if (lds_slices) {
        // ---- second split + one workgroup per 4096-entry slice of the table (index_kernels.h)
        const u64 nsub = (u64)NB * kSubs;
        if ((rc = buf_ensure_lend(ctx, SC_PART2, (ref->nbases + 1) * 8 + (nsub + 1) * 16 + nsub * 4 + 256, {SC_SEED_POOL, SC_SEED_CHAIN, SC_SEED_OUT})) ||
            (rc = buf_ensure(ctx, ctx->scratch[SC_SPLIT2], nsub * 4 + (size_t)(NB + 1) * 8 + (size_t)NB * 4 + 256))) return rc;
        char* pb = (char*)ctx->scratch[SC_PART2].p;
        d_part2 = (u64*)pb; pb += (ref->nbases + 1) * 8;
        d_sub = (u64*)pb; pb += (nsub + 1) * 8;
        d_bbase = (u64*)pb; pb += (nsub + 1) * 8;          // [NB + 1] used
        d_kept = (u32*)pb;
        char* qb = (char*)ctx->scratch[SC_SPLIT2].p;            // the same for the non-zero table entries
        d_cbase = (u64*)qb; qb += (size_t)(NB + 1) * 8;
        d_pres = (u32*)qb; qb += nsub * 4;
        d_bpres = (u32*)qb;
        if (g_split_threads == 512) hipLaunchKernelGGL(k_subpart<512>, dim3(NB), dim3(512), 0, s, (const u64*)d_part, (const u64*)d_bstart, NB, d_part2, d_sub);
        else hipLaunchKernelGGL(k_subpart<256>, dim3(NB), dim3(256), 0, s, (const u64*)d_part, (const u64*)d_bstart, NB, d_part2, d_sub);
        NECAT_CHECK_LAUNCH(ctx, "k_subpart");
        NECAT_HIP(ctx, hipMemsetAsync(d_bcnt, 0, (size_t)NB * 4, s));                // reused: kept entries per bucket
        NECAT_HIP(ctx, hipMemsetAsync(d_bpres, 0, (size_t)NB * 4, s));
        nsl = (b_hi - b_lo) * kSubs;                 // slices of this rank's hash range
        s0 = b_lo * kSubs;
        hipLaunchKernelGGL(k_slice_count, dim3(nsl), dim3(256), 0, s, (const u64*)d_part2, (const u64*)d_sub, (u32)max_occ, d_kept, d_bcnt, d_pres, d_bpres, s0);
        NECAT_CHECK_LAUNCH(ctx, "k_slice_count");
        hipLaunchKernelGGL(k_bucket_base, dim3(1), dim3(1024), 0, s, (const u32*)d_bcnt, NB, d_bbase);
        hipLaunchKernelGGL(k_bucket_base, dim3(1), dim3(1024), 0, s, (const u32*)d_bpres, NB, d_cbase);
        NECAT_CHECK_LAUNCH(ctx, "k_bucket_base");
        NECAT_HIP(ctx, hipMemcpyAsync(&mine[0], d_bbase + NB, 8, hipMemcpyDeviceToHost, s));
        NECAT_HIP(ctx, hipMemcpyAsync(&mine[1], d_cbase + NB, 8, hipMemcpyDeviceToHost, s));
        NECAT_HIP(ctx, hipStreamSynchronize(s));
    }
    return NECAT_OK;
    };   // local_phase
    rc = local_phase();
    uint64_t n_off = 0;
    if (lds_slices) {
        // the sizes of all ranks -> where this rank's entries sit in the gathered offset list / compact table; the third word is
        // this rank's status so far (a failed rank still takes part in the exchange: nobody waits for it in vain)
        const uint64_t n_local = mine[0];
        std::vector<unsigned long long> counts(2 * (size_t)G);
        counts[0] = mine[0]; counts[1] = mine[1];
        uint64_t base_add = 0, cbase_add = 0, n_comp = mine[1];
        n_off = mine[0];
        if (sharded) {
            unsigned long long msg[3] = {mine[0], mine[1], (unsigned long long)(unsigned)rc};
            std::vector<unsigned long long> all(3 * (size_t)G);
            const int rg = comm::host_allgather(ctx, comm, msg, all.data(), 24);
            if (rc) return rc;
            if (rg) return rg;
            for (int g = 0; g < G; ++g) if (all[3 * g + 2]) return set_err(ctx, NECAT_ERR_COMM, "rank %d failed in its slice of the index build (status %d)", g, (int)all[3 * g + 2]);
            n_off = 0; n_comp = 0;
            for (int g = 0; g < G; ++g) {
                counts[2 * g] = all[3 * g]; counts[2 * g + 1] = all[3 * g + 1];
                if (g < rk) { base_add += counts[2 * g]; cbase_add += counts[2 * g + 1]; }
                n_off += counts[2 * g]; n_comp += counts[2 * g + 1];
            }
            if (n_off >= (1ULL << 32)) return set_err(ctx, NECAT_ERR_INTERNAL, "ranks disagree on the volume (offset list of %llu entries)", (unsigned long long)n_off);
        } else if (rc) return rc;
        ix->n_offsets = n_off; ix->n_compact = n_comp;
        auto emit_phase = [&]() -> int {
        const size_t words_bytes = (size_t)(T / 64) * sizeof(IdxWord);
        if ((rc = table_alloc(ctx, ix, words_bytes + (n_comp + 1) * 8))) return rc;
        ix->words = ix->table; ix->compact = (uint64_t*)((char*)ix->table + words_bytes);
        // the offset list takes over the buffer of the fine buckets: k_subpart was their last reader, and a third array of 8 bytes per
        // base is what the first build of a process at oc2mkdb's 2 Gbp cut waited for (16 GB more device memory to map and clear).
        // Not in a sharded build: its offset list is published to the peers (HIP IPC), and with scratch buffers joining the pool of
        // published allocations `hipIpcGetMemHandle` failed with "invalid argument" in the second step of the two-rank pairs bench
        // (tests/test_gpu_pairs.py; profiles/NOTES_r04.md 6) - there the list keeps its own allocation as before.
        if (!sharded && ctx->scratch[SC_PART].p && ctx->scratch[SC_PART].cap >= (n_off + 1) * 8 && !(getenv("NECAT_INDEX_OWN_OFFSETS") && atoi(getenv("NECAT_INDEX_OWN_OFFSETS")))) {
            ix->offset_list = (uint64_t*)ctx->scratch[SC_PART].p; ix->offs_cap = ctx->scratch[SC_PART].cap; ctx->scratch[SC_PART] = DevBuf(); d_part = nullptr;
        }
        else if (ctx->idx_cache[1].p && ctx->idx_cache[1].cap >= (n_off + 1) * 8) { ix->offset_list = (uint64_t*)ctx->idx_cache[1].p; ix->offs_cap = ctx->idx_cache[1].cap; ctx->idx_cache[1] = DevBuf(); }
        else { NECAT_HIP(ctx, hipMalloc((void**)&ix->offset_list, (n_off + 1) * 8 + (n_off >> 4))); ix->offs_cap = (n_off + 1) * 8 + (n_off >> 4); }
        if ((rc = buf_ensure_lend(ctx, SC_TMPLIST, (n_local + 1) * 4, {SC_SEED_CHAIN, SC_SEED_OUT, SC_SEED_POOL}))) return rc;
        // 512 threads per slice: 4 workgroups (32 waves) per CU instead of 5 x 4 waves with 256 - the kernel is a chain of short
        // barrier-separated phases and needs the waves to hide their latencies (10.6 -> 9.8 ms for the whole build)
        // (slices of more than ~ 1000 records on average - volumes above 0.27 Gbp at k = 15 - rank in a bigger LDS buffer: index_kernels.h)
        const int emit_big = getenv("NECAT_INDEX_EMIT_BIG") ? atoi(getenv("NECAT_INDEX_EMIT_BIG")) : -1;          // (tests force either instance)
        const bool big = emit_big >= 0 ? emit_big != 0 : (nsl && n_local / nsl > 1000);
        if (big)
        hipLaunchKernelGGL((k_slice_emit<512, kLdsTmpBig>), dim3(nsl), dim3(512), 0, s, (const u64*)d_part2, (const u64*)d_sub, (u32)max_occ, (const u64*)d_bbase, (const u32*)d_kept,
                           (const u64*)d_cbase, (const u32*)d_pres, (IdxWord*)ix->words, ix->compact,
                           (u32*)ctx->scratch[SC_TMPLIST].p - base_add, ix->offset_list, s0, base_add, cbase_add);
        else
        hipLaunchKernelGGL((k_slice_emit<512, kLdsTmp>), dim3(nsl), dim3(512), 0, s, (const u64*)d_part2, (const u64*)d_sub, (u32)max_occ, (const u64*)d_bbase, (const u32*)d_kept,
                           (const u64*)d_cbase, (const u32*)d_pres, (IdxWord*)ix->words, ix->compact,
                           (u32*)ctx->scratch[SC_TMPLIST].p - base_add, ix->offset_list, s0, base_add, cbase_add);
        NECAT_CHECK_LAUNCH(ctx, "k_slice_emit");
        if (sharded) NECAT_HIP(ctx, hipEventRecord(ctx->ev[1], s));
        return NECAT_OK;
        };   // emit_phase
        rc = emit_phase();
        if (sharded) rc = comm::agree(ctx, comm, rc);            // every rank has its buffers and its slice under way, or nobody exchanges
        if (rc) return rc;
        if (sharded) {
            std::vector<comm::Part> pw(G), pc(G), po(G);
            uint64_t run = 0, crun = 0;
            for (int g = 0; g < G; ++g) {
                const u64 lo = (u64)g * NB / G, hi = (u64)(g + 1) * NB / G;
                pw[g].off = (size_t)((lo << pshift) / 64) * sizeof(IdxWord); pw[g].bytes = (size_t)(((hi - lo) << pshift) / 64) * sizeof(IdxWord);
                pc[g].off = (size_t)crun * 8; pc[g].bytes = (size_t)counts[2 * g + 1] * 8; crun += counts[2 * g + 1];
                po[g].off = (size_t)run * 8; po[g].bytes = (size_t)counts[2 * g] * 8; run += counts[2 * g];
            }
            for (auto* parts : {&pw, &pc, &po}) {
                void* basep = parts == &pw ? ix->words : parts == &pc ? (void*)ix->compact : (void*)ix->offset_list;
                if ((rc = comm::agree(ctx, comm, comm::allgatherv_inplace(ctx, comm, basep, *parts, s)))) return rc;
                ctx->shard_tm.index_exchange_ms += comm->last_ms; ctx->shard_tm.index_exchange_bytes += comm->last_bytes;
            }
            ctx->shard_tm.index_local_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
        }
    } else {
    if (rc) return rc;
    if (partitioned) {
        const u64 avg = ref->nbases / NB + 1;
        bchunks = (u32)std::max<u64>(1, (avg + kBucketChunk - 1) / kBucketChunk);
        hipLaunchKernelGGL(k_bucket_pass<0>, dim3(NB * bchunks), dim3(256), 0, s, (const u64*)d_part, (const u64*)d_bstart, NB, bchunks, cnt32, (u64)0, (u64*)nullptr);
        NECAT_CHECK_LAUNCH(ctx, "k_bucket_pass<count>");
    } else {
        hipLaunchKernelGGL(k_kmer_pass<0>, dim3(pass_grid), dim3(256), 0, s, vol, kmer_size, cnt32, (u64)0, (u64*)nullptr);
        NECAT_CHECK_LAUNCH(ctx, "k_kmer_pass<count>");
    }
    hipLaunchKernelGGL(k_tile_sums, dim3((unsigned)ntiles), dim3(256), 0, s, cnt32, T, (u32)max_occ, partial);
    NECAT_CHECK_LAUNCH(ctx, "k_tile_sums");
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, partial, ntiles);
    NECAT_CHECK_LAUNCH(ctx, "k_scan_partials");
    hipLaunchKernelGGL(k_write_stats, dim3((unsigned)ntiles), dim3(256), 0, s, cnt32, T, (u32)max_occ, partial, ix->kmer_stats);
    NECAT_CHECK_LAUNCH(ctx, "k_write_stats");
    NECAT_HIP(ctx, hipMemcpyAsync(&n_off, partial + ntiles, 8, hipMemcpyDeviceToHost, s));
    NECAT_HIP(ctx, hipStreamSynchronize(s));
    ix->n_offsets = n_off;
    if (ctx->idx_cache[1].p && ctx->idx_cache[1].cap >= (n_off + 1) * 8) { ix->offset_list = (uint64_t*)ctx->idx_cache[1].p; ix->offs_cap = ctx->idx_cache[1].cap; ctx->idx_cache[1] = DevBuf(); }
    else { NECAT_HIP(ctx, hipMalloc((void**)&ix->offset_list, (n_off + 1) * 8 + (n_off >> 4))); ix->offs_cap = (n_off + 1) * 8 + (n_off >> 4); }
    if (n_off) {
        if ((rc = buf_ensure(ctx, ctx->scratch[SC_TMPLIST], n_off * 8))) return rc;
        u64* tmp = (u64*)ctx->scratch[SC_TMPLIST].p;
        if (partitioned) {
            hipLaunchKernelGGL(k_bucket_pass<1>, dim3(NB * bchunks), dim3(256), 0, s, (const u64*)d_part, (const u64*)d_bstart, NB, bchunks, cnt32, n_off, tmp);
            NECAT_CHECK_LAUNCH(ctx, "k_bucket_pass<scatter>");
        } else {
            hipLaunchKernelGGL(k_kmer_pass<1>, dim3(pass_grid), dim3(256), 0, s, vol, kmer_size, cnt32, n_off, tmp);
            NECAT_CHECK_LAUNCH(ctx, "k_kmer_pass<scatter>");
        }
        hipLaunchKernelGGL(k_rank_buckets, dim3(grid_for(n_off, 256, 1u << 16)), dim3(256), 0, s, vol, kmer_size, (const u64*)ix->kmer_stats, (const u64*)tmp, n_off, ix->offset_list);
        NECAT_CHECK_LAUNCH(ctx, "k_rank_buckets");
    }
    }
    NECAT_HIP(ctx, hipEventRecord(ctx->ev[1], s));
    NECAT_HIP(ctx, hipStreamSynchronize(s));
    ctx->tm.index_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
    if (!sharded) ctx->shard_tm.index_local_ms = ctx->tm.index_ms;
    if (g_trace) fprintf(stderr, "[necat] index: events %.2f ms, host wall %.2f ms (local %.2f ms, exchange %.2f ms, %.1f MB received)\n", ctx->tm.index_ms, wall_ms() - w0,
                         ctx->shard_tm.index_local_ms, ctx->shard_tm.index_exchange_ms, ctx->shard_tm.index_exchange_bytes / 1e6);
    return NECAT_OK;
}

int index_build_impl(necat_ctx* ctx, necat_comm* comm, const necat_volume* ref, int kmer_size, int max_occ, necat_index** out)
{
    if (!ctx || !ref || !out) return NECAT_ERR_ARG;
    *out = nullptr;
    necat_index* ix = new necat_index();
    int rc = index_build_body(ctx, comm, ref, kmer_size, max_occ, ix);
    // a replicated build (every rank the whole table) has no collective step of its own: the ranks report their status here, so that a rank-local
    // failure (an allocation, a launch) fails the call on EVERY rank instead of leaving the peers waiting in the next collective (find / map_pair)
    // for a rank that has already returned.  (A sliced build agrees before each of its exchanges; a disagreement on the plan fails on every rank.)
    if (comm && comm->nranks > 1 && !ctx->shard_tm.index_sharded && rc != NECAT_ERR_COMM) rc = comm::agree(ctx, comm, rc);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); necat_index_free(ctx, ix); return rc; }
    *out = ix;
    return NECAT_OK;
}
}  // namespace

int necat_index_size(const necat_index* ix, uint64_t* table_entries, uint64_t* n_offsets)
{
    if (!ix) return NECAT_ERR_ARG;
    if (table_entries) *table_entries = ix->table_entries;
    if (n_offsets) *n_offsets = ix->n_offsets;
    return NECAT_OK;
}

int necat_index_download(necat_ctx* ctx, const necat_index* ix, uint64_t* kmer_stats, uint64_t* offset_list)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ix) return NECAT_ERR_ARG;
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    if (kmer_stats) {
        if (ix->kmer_stats) NECAT_HIP(ctx, hipMemcpy(kmer_stats, ix->kmer_stats, ix->table_entries * 8, hipMemcpyDeviceToHost));
        else {
            // the sparse table written out in the reference layout, a piece at a time (the dense table need not fit beside everything else)
            const uint64_t piece = std::min<uint64_t>(ix->table_entries, 1ULL << 27);
            if (int rc = buf_ensure(ctx, ctx->scratch[SC_CNT32], piece * 8)) return rc;
            u64* d = (u64*)ctx->scratch[SC_CNT32].p;
            IndexView v = index_view(ix);
            for (uint64_t h0 = 0; h0 < ix->table_entries; h0 += piece) {
                IndexView w = v; w.words = v.words + h0 / 64;       // lookup(h) of the shifted view = the entry h0 + h (h0 is a multiple of 64)
                hipLaunchKernelGGL(k_index_expand, dim3(grid_for(piece, 256, 1u << 16)), dim3(256), 0, ctx->stream, w, piece, d);
                NECAT_CHECK_LAUNCH(ctx, "k_index_expand");
                NECAT_HIP(ctx, hipMemcpyAsync(kmer_stats + h0, d, piece * 8, hipMemcpyDeviceToHost, ctx->stream));
                NECAT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            }
        }
    }
    if (offset_list && ix->n_offsets) NECAT_HIP(ctx, hipMemcpy(offset_list, ix->offset_list, ix->n_offsets * 8, hipMemcpyDeviceToHost));
    return NECAT_OK;
}

int necat_index_sparse_size(const necat_index* ix, uint64_t* n_pairs, uint64_t* n_compact)
{
    if (!ix) return NECAT_ERR_ARG;
    const bool sparse = ix->words != nullptr && ix->kmer_stats == nullptr;
    if (n_pairs) *n_pairs = sparse ? ix->table_entries / 64 : 0;
    if (n_compact) *n_compact = sparse ? ix->n_compact : 0;
    return NECAT_OK;
}

int necat_index_download_sparse(necat_ctx* ctx, const necat_index* ix, uint64_t* pairs, uint64_t* compact, uint64_t* offset_list)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ix) return NECAT_ERR_ARG;
    if (!ix->words || ix->kmer_stats) return set_err(ctx, NECAT_ERR_ARG, "the index holds the dense table (k = %d): use necat_index_download", ix->k);
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    if (pairs) NECAT_HIP(ctx, hipMemcpy(pairs, ix->words, (size_t)(ix->table_entries / 64) * sizeof(IdxWord), hipMemcpyDeviceToHost));
    if (compact && ix->n_compact) NECAT_HIP(ctx, hipMemcpy(compact, ix->compact, ix->n_compact * 8, hipMemcpyDeviceToHost));
    if (offset_list && ix->n_offsets) NECAT_HIP(ctx, hipMemcpy(offset_list, ix->offset_list, ix->n_offsets * 8, hipMemcpyDeviceToHost));
    return NECAT_OK;
}

void necat_index_free(necat_ctx* ctx, necat_index* ix)
{
    KnobScope knob_scope_(ctx);
    if (!ix) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    auto give = [&](void* p, size_t cap, DevBuf& slot) {
        if (!p) return;
        if (ctx && cap > slot.cap) { if (slot.p) (void)hipFree(slot.p); slot.p = p; slot.cap = cap; }
        else (void)hipFree(p);
    };
    DevBuf none;
    give(ix->table, ix->stats_cap, ctx ? ctx->idx_cache[0] : none);
    give(ix->offset_list, ix->offs_cap, ctx ? ctx->idx_cache[1] : none);
    delete ix;
}
