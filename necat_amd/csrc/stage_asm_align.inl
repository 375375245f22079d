// stage_asm_align.inl - oc2asmpm: the 2048-bp block aligner on the device.
// One of the stage files of libnecat_hip.so's single translation unit: necat_hip.hip includes them in order, inside its extern "C" block, after the
// context / knob / result-pool code they all use (the kernels are header templates and the stages share host helpers: one device code object, one 30 s build).

// ------------------------------------------------------------------------------------------ the block aligner of oc2asmpm

namespace {
// The cooperative path (asm_coop.h): every anchor an ExtTask, one block per task and round, the extension stage's kernels at the
// 2048-bp geometry.  h: validated anchors with local ids.
int asm_align_coop(necat_ctx* ctx, const necat_volume* ref, const necat_volume* reads, const std::vector<AsmAnchor>& h, double error, int min_align_size,
                   necat_alignment** aln, uint8_t** ops, uint64_t** ops_off)
{
    const uint64_t n = h.size();
    hipStream_t s = ctx->stream;
    const DevVolume drd = dev_view(reads), dref = dev_view(ref);
    // per-task column region: left stream (<= qoff + soff columns) then right stream (<= what is left of both reads), 2 bits per column
    std::vector<u64> base(n + 1, 0);
    for (uint64_t i = 0; i < n; ++i) {
        const u64 ql = reads->h_seq_off[h[i].q + 1] - reads->h_seq_off[h[i].q], sl = ref->h_seq_off[h[i].s + 1] - ref->h_seq_off[h[i].s];
        base[i + 1] = base[i] + ((ql + sl + (u64)h[i].qoff + (u64)h[i].soff + 64) / 32 + 2) * 8;
    }
    const u32 cap = (u32)((n + 63) & ~63ULL) + 64;          // capacity of every item array (list A is filled from both ends)
    const u32 groups = cap / 64 + 1;
    // band pools: a list runs in chunks of what its pool holds (as the 512-bp stage's capped pools), at least one group
    const size_t pool_cap = g_band_pool ? std::max<size_t>(g_band_pool, kAsmSlab) : (size_t)64 << 30;
    const u32 gchunkA = (u32)std::max<size_t>(1, std::min<size_t>(groups, pool_cap / kAsmSlabA));
    const u32 gchunkB = (u32)std::max<size_t>(1, std::min<size_t>(groups, pool_cap / kAsmSlab));
    int rc;
    const size_t misc = n * (sizeof(AsmAnchor) + sizeof(ExtTask) + 8) + (size_t)cap * 4 * sizeof(BlockItem) + (size_t)groups * 64 * 2 * sizeof(BlockResult) + (n + 1) * 8 + 8192;
    // checkpoint pool of the recompute path: per block 128 slots x 32 words x 16 B + 64 x 32 x 8 B of deltas = 80 KB (list A), 154 KB (list B)
    constexpr size_t kCkA = (size_t)RcGeom<kAsmBlock>::kCk * kAsmWordsA * 16, kHcA = (size_t)RcGeom<kAsmBlock>::kSeg * kAsmWordsA * 8;
    constexpr size_t kCkB = (size_t)RcGeom<kAsmCols>::kCk * kAsmWords * 16, kHcB = (size_t)RcGeom<kAsmCols>::kSeg * kAsmWords * 8;
    // (2 GB + 1 GB by default, NECAT_ASM_RC_POOL_MB: 26 k list-A / 6.8 k list-B blocks per launch still are 13 k / 6.8 k waves, and the 2 x 9 GB the
    // extension stage's cap allowed were most of what this short-lived program mapped - profiles/NOTES_r04.md 4)
    static const size_t asm_pool = (size_t)std::max<unsigned long long>(256, getenv("NECAT_ASM_RC_POOL_MB") ? strtoull(getenv("NECAT_ASM_RC_POOL_MB"), nullptr, 10) : 2048ULL) << 20;
    const u32 rc_chunkA = (u32)std::max<size_t>(64, std::min<size_t>((size_t)groups * 64, (asm_pool / (kCkA + kHcA)) & ~(size_t)63));
    const u32 rc_chunkB = (u32)std::max<size_t>(64, std::min<size_t>((size_t)groups * 64, ((asm_pool / 2) / (kCkB + kHcB)) & ~(size_t)63));
    // (the recompute path runs the two lists of a round side by side on two streams: list B has buffers of its own)
    if (g_asm_rc) {
        if ((rc = ext_streams(ctx)) ||
            (rc = buf_ensure(ctx, ctx->scratch[SC_EXT_CKPT], (size_t)rc_chunkA * (kCkA + kHcA))) ||
            (rc = buf_ensure(ctx, ctx->scratch[SC_EXT_CKPTB], (size_t)rc_chunkB * (kCkB + kHcB))) ||
            (rc = buf_ensure(ctx, ctx->scratch[SC_EXT_WOUT], (size_t)groups * 64 * sizeof(WalkOut))) ||
            (rc = buf_ensure(ctx, ctx->scratch[SC_EXT_WOUTB], (size_t)groups * 64 * sizeof(WalkOut)))) return rc;
    }
    ulonglong2* const rc_ck = (ulonglong2*)ctx->scratch[SC_EXT_CKPT].p;
    ulonglong2* const rc_ckB = (ulonglong2*)ctx->scratch[SC_EXT_CKPTB].p;
    u64* const rc_hcA = (u64*)((char*)ctx->scratch[SC_EXT_CKPT].p + (size_t)rc_chunkA * kCkA);
    u64* const rc_hcB = (u64*)((char*)ctx->scratch[SC_EXT_CKPTB].p + (size_t)rc_chunkB * kCkB);
    WalkOut* const d_wout = (WalkOut*)ctx->scratch[SC_EXT_WOUT].p;
    WalkOut* const d_woutB = (WalkOut*)ctx->scratch[SC_EXT_WOUTB].p;
    const size_t opsA_bytes = (size_t)groups * 64 * kAsmOpsA, opsB_bytes = (size_t)groups * 64 * kAsmMaxOps;
    const size_t fragA_bytes = (size_t)groups * 64 * kAsmFragWordsA * 8, fragB_bytes = (size_t)groups * 64 * kAsmFragWords * 8;
    if ((rc = g_asm_rc ? 0 : buf_ensure(ctx, ctx->scratch[SC_ASM_BAND], std::max((size_t)gchunkA * kAsmSlabA, (size_t)gchunkB * kAsmSlab))) ||
        (rc = buf_ensure(ctx, ctx->scratch[SC_ASM_OPS], g_asm_rc ? opsA_bytes + opsB_bytes : std::max(opsA_bytes, opsB_bytes))) ||
        (rc = buf_ensure(ctx, ctx->scratch[SC_ASM_FRAG], g_asm_rc ? fragA_bytes + fragB_bytes : std::max(fragA_bytes, fragB_bytes))) ||
        (rc = buf_ensure(ctx, ctx->scratch[SC_ASM_COLS], base[n] + 64)) ||
        (rc = buf_ensure(ctx, ctx->scratch[SC_ASM_MISC], misc))) return rc;
    char* mb = (char*)ctx->scratch[SC_ASM_MISC].p;
    auto take = [&](size_t bytes) { char* p = mb; mb += (bytes + 255) & ~(size_t)255; return p; };
    u32* d_count = (u32*)take(256);                    // [0..3] list buffer 0, [4..7] list buffer 1 (ExtLists counters: full A blocks, B blocks, other A blocks), [16] error flag, [32..] work counters
    int* d_err = (int*)(d_count + 16);
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_STATS], kStatBytes))) return rc;
    unsigned long long* d_stats = (unsigned long long*)ctx->scratch[SC_STATS].p;       // (work counters nobody reads here; the kernels want their kStatSlots copies)
    AsmAnchor* d_anchor = (AsmAnchor*)take(n * sizeof(AsmAnchor));
    ExtTask* d_tasks = (ExtTask*)take(n * sizeof(ExtTask));
    BlockItem* d_itemsA[2]; BlockItem* d_itemsB[2];
    for (int k = 0; k < 2; ++k) { d_itemsA[k] = (BlockItem*)take((size_t)cap * sizeof(BlockItem)); d_itemsB[k] = (BlockItem*)take((size_t)cap * sizeof(BlockItem)); }
    u64* d_base = (u64*)take((n + 1) * 8);
    BlockResult* d_res = (BlockResult*)take(((size_t)groups * 64) * sizeof(BlockResult));
    BlockResult* d_resB = (BlockResult*)take(((size_t)groups * 64) * sizeof(BlockResult));
    u8* d_cols = (u8*)ctx->scratch[SC_ASM_COLS].p;
    NECAT_HIP(ctx, hipMemcpyAsync(d_anchor, h.data(), n * sizeof(AsmAnchor), hipMemcpyHostToDevice, s));
    NECAT_HIP(ctx, hipMemcpyAsync(d_base, base.data(), (n + 1) * 8, hipMemcpyHostToDevice, s));
    NECAT_HIP(ctx, hipMemsetAsync(d_count, 0, 256, s));
    NECAT_HIP(ctx, hipEventRecord(ctx->ev[0], s));
    ctx->tm.myers_ms = ctx->tm.traceback_ms = 0; ctx->tm.myers_launches = ctx->tm.myers_blocks = ctx->tm.rounds = 0;
    auto lists = [&](int k) { ExtLists L; L.count = d_count + 4 * k; L.itemsA = d_itemsA[k]; L.itemsB = d_itemsB[k]; L.task_ops = d_cols; L.capA = cap; return L; };
    hipLaunchKernelGGL(k_asm_init, dim3(grid_for(n, 256)), dim3(256), 0, s, (const AsmAnchor*)d_anchor, (u32)n, (const u64*)reads->seq_off, (const u64*)ref->seq_off, d_tasks, lists(0),
                       (const u64*)d_base);
    NECAT_CHECK_LAUNCH(ctx, "k_asm_init");
    u64* const d_frag = (u64*)ctx->scratch[SC_ASM_FRAG].p;
    u8* const d_ops = (u8*)ctx->scratch[SC_ASM_OPS].p;
    u64* const d_fragB = g_asm_rc ? (u64*)((char*)d_frag + fragA_bytes) : d_frag;
    u8* const d_opsB = g_asm_rc ? d_ops + opsA_bytes : d_ops;
    hipStream_t sB = g_asm_rc ? ctx->stream_b : s;
    for (u32 r = 0;; ++r) {
        if (r > 4096) return set_err(ctx, NECAT_ERR_INTERNAL, "asm aligner: no end of rounds");
        const int cur = (int)(r & 1), nxt = cur ^ 1;
        u32 cnt[4] = {0, 0, 0, 0};
        NECAT_HIP(ctx, hipMemcpyAsync(cnt, d_count + 4 * cur, 16, hipMemcpyDeviceToHost, s));
        NECAT_HIP(ctx, hipStreamSynchronize(s));
        const u32 nf = cnt[0], nB = cnt[1], np = cnt[2];
        if (nf + nB + np == 0) break;
        NECAT_HIP(ctx, hipMemsetAsync(d_count + 4 * nxt, 0, 16, s));
        if (g_asm_rc) { NECAT_HIP(ctx, hipEventRecord(ctx->ev[34], s)); NECAT_HIP(ctx, hipStreamWaitEvent(sB, ctx->ev[34], 0)); }
        const ExtLists next = lists(nxt);
        RoundCtl ctl;
        double dp = 0, wk = 0;
        // ---- list A: work indices [0, nf) the full blocks, [nf16, nf16 + np) the others (ListView)
        const u32 boundA = (nf + np) ? ((nf + 15u) & ~15u) + np : 0u;
        if (boundA) {
            const u32 gA = (boundA + 63) / 64;
            const u32* d_nA = d_count + 4 * cur;
            hipLaunchKernelGGL((k_ext_frag<kAsmWordsA, kAsmTWordsA>), dim3(grid_for((u64)gA * 64 * kFragSplit, 256)), dim3(256), 0, s,
                               drd, dref, (const BlockItem*)d_itemsA[cur], boundA, d_nA, cap, d_frag, ctl);
            NECAT_CHECK_LAUNCH(ctx, "k_ext_frag<asm A>");
            if (g_asm_rc) {
                // SHW pass with checkpoints + deltas, then the walk that recomputes the two words it stands on (ext_rcwalk.h), chunk by chunk
                // through the checkpoint buffer; then one finishing launch for the whole list
                const u32 epoch = ++ctx->epoch & 0x3fffffu, fl = epoch | (1u << 27);
                NECAT_HIP(ctx, hipEventRecord(ctx->ev[2], s));
                for (u32 lo = 0; lo < boundA; lo += rc_chunkA) {
                    const u32 hi = std::min<u64>((u64)lo + rc_chunkA, (u64)gA * 64), cn = hi - lo;
                    hipLaunchKernelGGL((k_myers_ckg<kAsmWordsA, kAsmTWordsA, kAsmBlock, 32>), dim3((cn + 1) / 2), dim3(64), 0, s, (const BlockItem*)d_itemsA[cur], boundA, d_nA, cap,
                                       (const u64*)d_frag, rc_ck, rc_hcA, error, d_res, d_stats, epoch, lo, hi);
                    launch_rcwalk2<kAsmWordsA, kAsmTWordsA, kAsmBlock, kAsmOpsA>(cn, s, (const BlockItem*)d_itemsA[cur], boundA, d_nA, cap,
                                       (const u64*)d_frag, (const ulonglong2*)rc_ck, (const u64*)rc_hcA, (const BlockResult*)d_res, (const ExtTask*)d_tasks, 1, 8, d_ops, d_wout, d_stats, d_err, fl, lo, hi);
                    NECAT_CHECK_LAUNCH(ctx, "k_myers_ckg / k_rcwalk2<asm A>");
                }
                NECAT_HIP(ctx, hipEventRecord(ctx->ev[3], s));
                hipLaunchKernelGGL((k_traceback<kAsmWordsA, kAsmTWordsA, kAsmBlock, kAsmOpsA, false, 5, kAsmBlock, false, 4>), dim3((gA + 3) / 4), dim3(256), 0, s,
                                   (const BlockItem*)d_itemsA[cur], boundA, d_nA, cap, (const u64*)d_frag, (const char*)nullptr, (size_t)0,
                                   (const BlockResult*)d_res, d_ops, d_tasks, 8 /* kMatchCnt2: the tail match length of hbn_align */, (i32*)nullptr, d_err, next, fl, 0u, (const WalkOut*)d_wout);
                NECAT_CHECK_LAUNCH(ctx, "k_traceback<asm A, rc>");
                NECAT_HIP(ctx, hipEventRecord(ctx->ev[24], s));
                ctx->tm.myers_launches += 1;
            }
#if !NECAT_XCHECK
            else NECAT_RETIRED(ctx, "the 2048-bp blocks through k_myers_coop + band records (NECAT_ASM_RC=0)");
#else
            else
            for (u32 g0 = 0; g0 < gA; g0 += gchunkA) {
                const u32 lo = g0 * 64, hi = std::min(gA, g0 + gchunkA) * 64, cn = hi - lo;
                char* slabs = (char*)ctx->scratch[SC_ASM_BAND].p - (size_t)g0 * kAsmSlabA;         // the kernels index slabs by work index / 64
                const u32 epoch = ++ctx->epoch & 0x3fffffu;
                NECAT_HIP(ctx, hipEventRecord(ctx->ev[2], s));
                hipLaunchKernelGGL((k_myers_coop<kAsmWordsA, kAsmTWordsA, kAsmBlock, 32>), dim3(cn / 2), dim3(64), 0, s, (const BlockItem*)d_itemsA[cur], boundA, d_nA, cap,
                                   (const u64*)d_frag, slabs, kAsmSlabA, error, d_res, d_stats, epoch, lo);
                NECAT_CHECK_LAUNCH(ctx, "k_myers_coop<asm A>");
                NECAT_HIP(ctx, hipEventRecord(ctx->ev[3], s));
                if (g_walk_wave)
                    hipLaunchKernelGGL((k_walk_wave<kAsmWordsA, kAsmTWordsA, kAsmOpsA, kAsmBlock>), dim3(cn), dim3(64), 0, s, (const BlockItem*)d_itemsA[cur], boundA, d_nA, cap,
                                       (const u64*)d_frag, (const char*)slabs, kAsmSlabA, (const BlockResult*)d_res, d_tasks, 8 /* kMatchCnt2: the tail match length of hbn_align */, d_err, next, lo);
                else
                hipLaunchKernelGGL((k_traceback<kAsmWordsA, kAsmTWordsA, kAsmBlock, kAsmOpsA, false, 0, kAsmBlock>), dim3(cn / 64), dim3(64), 0, s,
                                   (const BlockItem*)d_itemsA[cur], boundA, d_nA, cap, (const u64*)d_frag, (const char*)slabs, kAsmSlabA,
                                   (const BlockResult*)d_res, d_ops, d_tasks, 8 /* kMatchCnt2: the tail match length of hbn_align */, (i32*)nullptr, d_err, next, epoch, lo);
                NECAT_CHECK_LAUNCH(ctx, "k_traceback<asm A>");
                NECAT_HIP(ctx, hipEventRecord(ctx->ev[24], s));
                NECAT_HIP(ctx, hipStreamSynchronize(s));
                dp += ev_ms(ctx->ev[2], ctx->ev[3]); wk += ev_ms(ctx->ev[3], ctx->ev[24]);
                ctx->tm.myers_launches += 1;
            }
#endif
        }
        // ---- list B: a plain list of nB items
        if (nB) {
            const u32 gB = (nB + 63) / 64;
            hipLaunchKernelGGL((k_ext_frag<kAsmWords, kAsmTWords>), dim3(grid_for((u64)gB * 64 * kFragSplit, 256)), dim3(256), 0, sB,
                               drd, dref, (const BlockItem*)d_itemsB[cur], nB, (const u32*)nullptr, 0u, d_fragB, ctl);
            NECAT_CHECK_LAUNCH(ctx, "k_ext_frag<asm B>");
            if (g_asm_rc) {
                const u32 epoch = ++ctx->epoch & 0x3fffffu, fl = epoch | (1u << 27);
                NECAT_HIP(ctx, hipEventRecord(ctx->ev[36], sB));
                for (u32 lo = 0; lo < nB; lo += rc_chunkB) {
                    const u32 hi = std::min<u64>((u64)lo + rc_chunkB, (u64)gB * 64), cn = std::min(hi, nB) - lo;
                    hipLaunchKernelGGL((k_myers_ckg<kAsmWords, kAsmTWords, kAsmCols, 64>), dim3(cn), dim3(64), 0, sB, (const BlockItem*)d_itemsB[cur], nB, (const u32*)nullptr, 0u,
                                       (const u64*)d_fragB, rc_ckB, rc_hcB, error, d_resB, d_stats, epoch, lo, hi);
                    launch_rcwalk2<kAsmWords, kAsmTWords, kAsmCols, kAsmMaxOps>(cn, sB, (const BlockItem*)d_itemsB[cur], nB, (const u32*)nullptr, 0u,
                                       (const u64*)d_fragB, (const ulonglong2*)rc_ckB, (const u64*)rc_hcB, (const BlockResult*)d_resB, (const ExtTask*)d_tasks, 1, 8, d_opsB, d_woutB, d_stats, d_err, fl, lo, hi);
                    NECAT_CHECK_LAUNCH(ctx, "k_myers_ckg / k_rcwalk2<asm B>");
                }
                NECAT_HIP(ctx, hipEventRecord(ctx->ev[37], sB));
                hipLaunchKernelGGL((k_traceback<kAsmWords, kAsmTWords, kAsmCols, kAsmMaxOps, false, 5, kAsmBlock, false, 4>), dim3((gB + 3) / 4), dim3(256), 0, sB,
                                   (const BlockItem*)d_itemsB[cur], nB, (const u32*)nullptr, 0u, (const u64*)d_fragB, (const char*)nullptr, (size_t)0,
                                   (const BlockResult*)d_resB, d_opsB, d_tasks, 8, (i32*)nullptr, d_err, next, fl, 0u, (const WalkOut*)d_woutB);
                NECAT_CHECK_LAUNCH(ctx, "k_traceback<asm B, rc>");
                NECAT_HIP(ctx, hipEventRecord(ctx->ev[38], sB));
                ctx->tm.myers_launches += 1;
            }
#if !NECAT_XCHECK
            else NECAT_RETIRED(ctx, "the 2048-bp blocks through k_myers_coop + band records (NECAT_ASM_RC=0)");
#else
            else
            for (u32 g0 = 0; g0 < gB; g0 += gchunkB) {
                const u32 lo = g0 * 64, hi = std::min(nB, (g0 + gchunkB) * 64), cn = hi - lo;
                char* slabs = (char*)ctx->scratch[SC_ASM_BAND].p - (size_t)g0 * kAsmSlab;
                const u32 epoch = ++ctx->epoch & 0x3fffffu;
                NECAT_HIP(ctx, hipEventRecord(ctx->ev[2], s));
                hipLaunchKernelGGL((k_myers_coop<kAsmWords, kAsmTWords, kAsmCols, 64>), dim3(cn), dim3(64), 0, s, (const BlockItem*)d_itemsB[cur], hi, (const u32*)nullptr, 0u,
                                   (const u64*)d_frag, slabs, kAsmSlab, error, d_res, d_stats, epoch, lo);
                NECAT_CHECK_LAUNCH(ctx, "k_myers_coop<asm B>");
                NECAT_HIP(ctx, hipEventRecord(ctx->ev[3], s));
                if (g_walk_wave)
                    hipLaunchKernelGGL((k_walk_wave<kAsmWords, kAsmTWords, kAsmMaxOps, kAsmBlock>), dim3(cn), dim3(64), 0, s, (const BlockItem*)d_itemsB[cur], hi, (const u32*)nullptr, 0u,
                                       (const u64*)d_frag, (const char*)slabs, kAsmSlab, (const BlockResult*)d_res, d_tasks, 8, d_err, next, lo);
                else
                hipLaunchKernelGGL((k_traceback<kAsmWords, kAsmTWords, kAsmCols, kAsmMaxOps, false, 0, kAsmBlock>), dim3((cn + 63) / 64), dim3(64), 0, s,
                                   (const BlockItem*)d_itemsB[cur], hi, (const u32*)nullptr, 0u, (const u64*)d_frag, (const char*)slabs, kAsmSlab,
                                   (const BlockResult*)d_res, d_ops, d_tasks, 8, (i32*)nullptr, d_err, next, epoch, lo);
                NECAT_CHECK_LAUNCH(ctx, "k_traceback<asm B>");
                NECAT_HIP(ctx, hipEventRecord(ctx->ev[24], s));
                NECAT_HIP(ctx, hipStreamSynchronize(s));
                dp += ev_ms(ctx->ev[2], ctx->ev[3]); wk += ev_ms(ctx->ev[3], ctx->ev[24]);
                ctx->tm.myers_launches += 1;
            }
#endif
        }
        if (g_asm_rc) {
            if (nB) { NECAT_HIP(ctx, hipEventRecord(ctx->ev[35], sB)); NECAT_HIP(ctx, hipStreamWaitEvent(s, ctx->ev[35], 0)); }
            NECAT_HIP(ctx, hipStreamSynchronize(s));
            if (boundA) { dp += ev_ms(ctx->ev[2], ctx->ev[3]); wk += ev_ms(ctx->ev[3], ctx->ev[24]); }
            if (nB) { dp += ev_ms(ctx->ev[36], ctx->ev[37]); wk += ev_ms(ctx->ev[37], ctx->ev[38]); }      // (the two chains overlap: the sums exceed the round's wall time)
        }
        ctx->tm.myers_ms += dp; ctx->tm.traceback_ms += wk;
        ctx->tm.myers_blocks += nf + np + nB; ctx->tm.rounds += 1;
        if (g_trace & 1) fprintf(stderr, "[necat] asm round %u: list A %u full + %u other blocks, list B %u blocks: DP %.3f ms, walk %.3f ms\n", r, nf, np, nB, dp, wk);
    }
    // results: coordinates + identity per anchor, the alignment columns packed in anchor order (as necat_onc_align_batch)
    const size_t out_fixed = n * (sizeof(necat_alignment) + 4 + 8) + 1024;
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_ASM_OUT], out_fixed))) return rc;
    char* ob = (char*)ctx->scratch[SC_ASM_OUT].p;
    necat_alignment* d_aln = (necat_alignment*)ob; ob += (n * sizeof(necat_alignment) + 255) & ~(size_t)255;
    u32* d_len = (u32*)ob; ob += (n * 4 + 255) & ~(size_t)255;
    u64* d_off = (u64*)ob;
    hipLaunchKernelGGL(k_ext_alignment, dim3(grid_for(n, 256)), dim3(256), 0, s, (const ExtTask*)d_tasks, (u32)n, 0u, min_align_size, d_aln, d_len);
    NECAT_CHECK_LAUNCH(ctx, "k_ext_alignment");
    necat_alignment* res = (necat_alignment*)result_alloc(n * sizeof(necat_alignment));
    uint64_t* off = (uint64_t*)result_alloc((n + 1) * 8);
    std::vector<u32> len(n);
    int herr = 0;
    auto fail = [&](int code) { necat_free(res); necat_free(off); return code; };
    if (!res || !off) return fail(set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"));
    if (hipMemcpyAsync(len.data(), d_len, n * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipMemcpyAsync(res, d_aln, n * sizeof(necat_alignment), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return fail(set_err(ctx, NECAT_ERR_DEVICE, "asm aligner: result copy failed: %s", hipGetErrorString(hipGetLastError())));
    if (herr) return fail(set_err(ctx, NECAT_ERR_INTERNAL, "asm aligner: the kernels reported error code %d", herr));
    // every alignment starts on a 64-bit word: 32 columns per word (offsets in bytes)
    std::vector<u64> woff(n + 1, 0);
    for (uint64_t i = 0; i < n; ++i) woff[i + 1] = woff[i] + (len[i] + 31) / 32;
    for (uint64_t i = 0; i <= n; ++i) off[i] = woff[i] * 8;
    const u64 tot = woff[n] * 8;
    uint8_t* packed = (uint8_t*)result_alloc(std::max<u64>(8, tot));
    if (!packed) return fail(set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"));
    if (tot) {
        if ((rc = buf_ensure(ctx, ctx->scratch[SC_EXT_COLS_OUT], tot + 64))) { necat_free(packed); return fail(rc); }
        hipError_t e = hipMemcpyAsync(d_off, woff.data(), n * 8, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(k_ext_strings, dim3(grid_for((u64)n * 64, 256)), dim3(256), 0, s, (const ExtTask*)d_tasks, (u32)n, (const u8*)d_cols, (const u64*)d_off, (u64*)ctx->scratch[SC_EXT_COLS_OUT].p);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(packed, ctx->scratch[SC_EXT_COLS_OUT].p, tot, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipEventRecord(ctx->ev[1], s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { necat_free(packed); return fail(set_err(ctx, NECAT_ERR_DEVICE, "asm aligner: column copy failed: %s", hipGetErrorString(e))); }
    } else { (void)hipEventRecord(ctx->ev[1], s); (void)hipStreamSynchronize(s); }
    ctx->tm.extend_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
    if (g_trace & 2) fprintf(stderr, "[necat] asm_align (cooperative): %lu anchors, %lu rounds, %lu blocks, DP %.2f ms, walk %.2f ms, whole call %.2f ms\n", (unsigned long)n,
                             (unsigned long)ctx->tm.rounds, (unsigned long)ctx->tm.myers_blocks, ctx->tm.myers_ms, ctx->tm.traceback_ms, ctx->tm.extend_ms);
    *aln = res; *ops = packed; *ops_off = off;
    return NECAT_OK;
}
}  // namespace

int necat_asm_align_batch(necat_ctx* ctx, const necat_volume* ref, const necat_volume* reads, int read_start_id, int ref_start_id,
                          const necat_asm_anchor* anchors, uint64_t n, double error, int min_align_size,
                          necat_alignment** aln, uint8_t** ops, uint64_t** ops_off)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ref || !reads || !aln || !ops || !ops_off || (n && !anchors)) return NECAT_ERR_ARG;
    *aln = nullptr; *ops = nullptr; *ops_off = nullptr;
    if (!(error > 0.0 && error <= 1.0) || n >= (1ULL << 31)) return set_err(ctx, NECAT_ERR_ARG, "error rate / count out of range");
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<AsmAnchor> h(n);
    std::vector<u64> coff(n + 1, 0);
    for (uint64_t i = 0; i < n; ++i) {
        const necat_asm_anchor& a = anchors[i];
        const int64_t lq = (int64_t)a.qid - read_start_id, ls = (int64_t)a.sid - ref_start_id;
        if (lq < 0 || (uint64_t)lq >= reads->nseq || ls < 0 || (uint64_t)ls >= ref->nseq || (a.sdir != 0 && a.sdir != 1))
            return set_err(ctx, NECAT_ERR_ARG, "anchor %lu refers to a read outside the volumes", (unsigned long)i);
        const u64 ql = reads->h_seq_off[lq + 1] - reads->h_seq_off[lq], sl = ref->h_seq_off[ls + 1] - ref->h_seq_off[ls];
        if (a.qoff < 0 || (u64)a.qoff > ql || a.soff < 0 || (u64)a.soff > sl || ql >= (1ULL << 31) || sl >= (1ULL << 31))
            return set_err(ctx, NECAT_ERR_ARG, "anchor %lu lies outside its reads", (unsigned long)i);
        h[i].q = (i32)lq; h[i].s = (i32)ls; h[i].sdir = a.sdir; h[i].qoff = a.qoff; h[i].soff = a.soff;
        coff[i + 1] = coff[i] + ((ql + sl + 64 + 7) & ~7ULL);          // a column consumes at least one base of one of the two
    }
    if (n && !g_asm_lane) return asm_align_coop(ctx, ref, reads, h, error, min_align_size, aln, ops, ops_off);
    necat_alignment* res = (necat_alignment*)result_alloc(std::max<uint64_t>(1, n) * sizeof(necat_alignment));
    uint64_t* off = (uint64_t*)result_alloc((n + 1) * 8);
    if (!res || !off) { necat_free(res); necat_free(off); return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"); }
    off[0] = 0;
    auto fail = [&](int rc) { necat_free(res); necat_free(off); return rc; };
    if (n == 0) { *aln = res; *ops_off = off; *ops = (uint8_t*)result_alloc(8); return NECAT_OK; }
#if !NECAT_XCHECK
    return fail(necat::set_err(ctx, NECAT_ERR_ARG, "the lane-per-alignment kernel k_asm_align (NECAT_ASM_LANE=1): a cross-check path this library is built without (libnecat_hip_xcheck.so)"));
#else
    // waves per launch: one band slab (126 MB) per wave inside the band-pool cap
    const size_t pool = g_band_pool ? std::max<size_t>(g_band_pool, kAsmBandWave) : (size_t)32 << 30;
    const u32 waves_total = (u32)((n + 63) / 64);
    const u32 waves_max = (u32)std::max<size_t>(1, std::min<size_t>(pool / kAsmBandWave, waves_total));
    int rc;
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_ASM_BAND], (size_t)waves_max * kAsmBandWave)) ||
        (rc = buf_ensure(ctx, ctx->scratch[SC_ASM_OPS], (size_t)waves_max * kAsmOpsWave)) ||
        (rc = buf_ensure(ctx, ctx->scratch[SC_ASM_COLS], coff[n] + 64)) ||
        (rc = buf_ensure(ctx, ctx->scratch[SC_ASM_MISC], n * (sizeof(AsmAnchor) + sizeof(AsmOut) + 8) + 1024))) return fail(rc);
    char* mb = (char*)ctx->scratch[SC_ASM_MISC].p;
    u64* d_coff = (u64*)mb; mb += ((n + 1) * 8 + 63) & ~63ULL;
    AsmOut* d_out = (AsmOut*)mb; mb += (n * sizeof(AsmOut) + 63) & ~63ULL;
    AsmAnchor* d_anchor = (AsmAnchor*)mb;
    u8* d_cols = (u8*)ctx->scratch[SC_ASM_COLS].p;
    if (hipMemcpyAsync(d_anchor, h.data(), n * sizeof(AsmAnchor), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_coff, coff.data(), (n + 1) * 8, hipMemcpyHostToDevice, s) != hipSuccess) return fail(set_err(ctx, NECAT_ERR_DEVICE, "anchor upload failed"));
    const DevVolume drd = dev_view(reads), dref = dev_view(ref);
    (void)hipEventRecord(ctx->ev[0], s);
    for (u32 w0 = 0; w0 < waves_total; w0 += waves_max) {
        const u32 nw = std::min(waves_max, waves_total - w0);
        const u64 first = (u64)w0 * 64, cnt = std::min<u64>((u64)nw * 64, n - first);
        hipLaunchKernelGGL(k_asm_align, dim3(nw), dim3(64), 0, s, (const AsmAnchor*)(d_anchor + first), (u32)cnt, drd, dref, error, 8 /* kMatchCnt2 */,
                           (char*)ctx->scratch[SC_ASM_BAND].p, (u8*)ctx->scratch[SC_ASM_OPS].p, d_cols, (const u64*)(d_coff + first), d_out + first);
        if (hipGetLastError() != hipSuccess) return fail(set_err(ctx, NECAT_ERR_DEVICE, "k_asm_align launch failed"));
    }
    std::vector<AsmOut> ho(n);
    std::vector<u8> hc(coff[n] + 8);
    (void)hipEventRecord(ctx->ev[1], s);
    if (hipMemcpyAsync(ho.data(), d_out, n * sizeof(AsmOut), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(hc.data(), d_cols, coff[n], hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) return fail(set_err(ctx, NECAT_ERR_DEVICE, "k_asm_align failed: %s", hipGetErrorString(hipGetLastError())));
    ctx->tm.extend_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
    // the alignment of an anchor: its left stream [lfrom, lto) read backwards, then its right stream [lto + rfrom, lto + rto); packed two bits per
    // column, every alignment on an 8-byte boundary
    for (uint64_t i = 0; i < n; ++i) {
        if (ho[i].err) return fail(set_err(ctx, NECAT_ERR_INTERNAL, "k_asm_align: anchor %lu reported error code %d", (unsigned long)i, ho[i].err));
        const int nl = ho[i].lto - ho[i].lfrom, nr = ho[i].rto - ho[i].rfrom;
        if (nl < 0 || nr < 0 || nl + nr != ho[i].cols) return fail(set_err(ctx, NECAT_ERR_INTERNAL, "k_asm_align: anchor %lu has inconsistent streams", (unsigned long)i));
        off[i + 1] = off[i] + (((uint64_t)(nl + nr) + 3) / 4 + 7 & ~7ULL);
    }
    uint8_t* packed = (uint8_t*)result_alloc(std::max<uint64_t>(8, off[n]));
    if (!packed) return fail(set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"));
    memset(packed, 0, std::max<uint64_t>(8, off[n]));
    cns::parallel_for(n, [&](size_t i) {
        const AsmOut& o = ho[i];
        const u8* c = hc.data() + coff[i];
        uint8_t* dst = packed + off[i];
        const int nl = o.lto - o.lfrom, nr = o.rto - o.rfrom;
        for (int j = 0; j < nl + nr; ++j) {
            const u8 op = j < nl ? c[o.lto - 1 - j] : c[o.lto + o.rfrom + (j - nl)];
            dst[j >> 2] |= (uint8_t)((op & 3) << (2 * (j & 3)));
        }
        necat_alignment& a = res[i];
        a.ok = o.cols >= min_align_size ? 1 : 0;
        a.qoff = o.qoff; a.qend = o.qend; a.toff = o.toff; a.tend = o.tend; a.align_size = o.cols;
        a.ident_perc = o.cols ? 100.0 * (double)o.mat / (double)o.cols : 0.0;
    });
    if (g_trace & 2) fprintf(stderr, "[necat] asm_align: %lu anchors, %u waves (%u per launch), kernels %.2f ms\n", (unsigned long)n, waves_total, waves_max, ctx->tm.extend_ms);
    *aln = res; *ops = packed; *ops_off = off;
    return NECAT_OK;
#endif
}
