// stage_edlib_batch.inl - batch Edlib_align (test / profiling hook).
// One of the stage files of libnecat_hip.so's single translation unit: necat_hip.hip includes them in order, inside its extern "C" block, after the
// context / knob / result-pool code they all use (the kernels are header templates and the stages share host helpers: one device code object, one 30 s build).

// ------------------------------------------------------------------------------------------ batch Edlib_align (test / profiling hook)

int necat_edlib_align_batch(necat_ctx* ctx, const uint8_t* seqs, uint64_t seqs_len, const uint64_t* q_off, const int32_t* q_len,
                            const uint64_t* t_off, const int32_t* t_len, uint64_t n, double error,
                            int32_t* dist, int32_t* qend, int32_t* tend, uint8_t** ops, uint64_t** ops_off)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !seqs || !q_off || !q_len || !t_off || !t_len || !dist || !qend || !tend) return NECAT_ERR_ARG;
    if (ops) *ops = nullptr;
    if (ops_off) *ops_off = nullptr;
    if (n == 0) return NECAT_OK;
#if !NECAT_XCHECK
    (void)seqs_len; (void)error;
    NECAT_RETIRED(ctx, "necat_edlib_align_batch (the block-by-block hook of the parity tests)");
#else
    for (uint64_t i = 0; i < n; ++i) {
        if (q_len[i] < 1 || t_len[i] < 1 || q_len[i] > kMaxFragLen || t_len[i] > kMaxFragLen ||
            q_off[i] + q_len[i] > seqs_len || t_off[i] + t_len[i] > seqs_len)
            return set_err(ctx, NECAT_ERR_ARG, "block %lu: fragment lengths must be 1..%d and inside seqs", (unsigned long)i, kMaxFragLen);
    }
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // pack to NECAT pac and upload as a one-read pseudo volume
    std::vector<uint8_t> pac((seqs_len + 3) / 4 + 8, 0);
    for (uint64_t i = 0; i < seqs_len; ++i) pac[i >> 2] |= (uint8_t)((seqs[i] & 3) << ((~i & 3) << 1));
    uint64_t off0 = 0, size0 = seqs_len;
    necat_volume* vol = nullptr;
    int rc = necat_volume_upload(ctx, pac.data(), seqs_len, &off0, &size0, 1, &vol);
    if (rc) return rc;
    DevVolume dv = dev_view(vol);
    // split into the two kernel shapes
    std::vector<BlockItem> itA, itB; std::vector<u64> idA, idB;
    for (uint64_t i = 0; i < n; ++i) {
        BlockItem it; it.g.q_base = (i64)q_off[i]; it.g.q_dir = 1; it.g.q_comp = 0; it.g.t_base = (i64)t_off[i]; it.g.t_dir = 1; it.g.t_comp = 0;
        it.task = -1; it.qn = (i16)q_len[i]; it.tn = (i16)t_len[i];
        if (q_len[i] == kOcaBlockSize && t_len[i] == kOcaBlockSize) { itA.push_back(it); idA.push_back(i); } else { itB.push_back(it); idB.push_back(i); }
    }
    ctx->tm.myers_ms = 0; ctx->tm.traceback_ms = 0; ctx->tm.myers_launches = 0; ctx->tm.myers_blocks = n; ctx->tm.myers_word_updates = 0;
    ctx->tm.myers_cells_bases = 0;
    std::vector<std::vector<uint8_t>> fwd_ops(n);
    int* d_err = nullptr;
    NECAT_HIP(ctx, hipMalloc((void**)&d_err, 4 + 4 + 24));
    NECAT_HIP(ctx, hipMemsetAsync(d_err, 0, 32, s));
    { const int rcs = buf_ensure(ctx, ctx->scratch[SC_STATS], kStatBytes); if (rcs) { (void)hipFree(d_err); return rcs; } }
    unsigned long long* d_stats = (unsigned long long*)ctx->scratch[SC_STATS].p;
    const u32 chunk = getenv("NECAT_BATCH_CHUNK") ? (u32)strtoul(getenv("NECAT_BATCH_CHUNK"), nullptr, 10) : 65536u;
    auto run_shape = [&](std::vector<BlockItem>& items, std::vector<u64>& ids, bool full) -> int {
        for (size_t base = 0; base < items.size(); base += chunk) {
            const u32 m = (u32)std::min<size_t>(chunk, items.size() - base);
            const u32 g = (m + 63) / 64;
            const size_t slab = full ? kSlabA : kSlabB;
            const int fw = full ? kFragWordsA : kFragWordsB, maxops = full ? kOpsA : kOpsB;
            int rc2;
            if ((rc2 = buf_ensure(ctx, ctx->scratch[SC_EXT_LISTS], (size_t)m * sizeof(BlockItem))) ||
                (rc2 = buf_ensure(ctx, ctx->scratch[SC_EXT_FRAG], (size_t)g * 64 * fw * 8)) ||
                (rc2 = ensure_zeroed(ctx, ctx->scratch[SC_EXT_MAT], (size_t)g * slab, s)) ||
                (rc2 = buf_ensure(ctx, ctx->scratch[SC_EXT_OPS], (size_t)g * 64 * maxops)) ||
                (rc2 = buf_ensure(ctx, ctx->scratch[SC_EXT_RES], (size_t)g * 64 * (sizeof(BlockResult) + 4)))) return rc2;
            BlockItem* d_items = (BlockItem*)ctx->scratch[SC_EXT_LISTS].p;
            u64* d_frag = (u64*)ctx->scratch[SC_EXT_FRAG].p;
            char* d_slabs = (char*)ctx->scratch[SC_EXT_MAT].p;
            u8* d_ops = (u8*)ctx->scratch[SC_EXT_OPS].p;
            BlockResult* d_res = (BlockResult*)ctx->scratch[SC_EXT_RES].p;
            i32* d_nops = (i32*)(d_res + (size_t)g * 64);
            NECAT_HIP(ctx, hipMemcpyAsync(d_items, items.data() + base, (size_t)m * sizeof(BlockItem), hipMemcpyHostToDevice, s));
            if (full) hipLaunchKernelGGL((k_ext_frag<kWordsA, kTWordsA>), dim3(grid_for((u64)g * 64 * kFragSplit, 256)), dim3(256), 0, s, dv, dv, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, d_frag, RoundCtl());
            else hipLaunchKernelGGL((k_ext_frag<kWordsB, kTWordsB>), dim3(grid_for((u64)g * 64 * kFragSplit, 256)), dim3(256), 0, s, dv, dv, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, d_frag, RoundCtl());
            NECAT_CHECK_LAUNCH(ctx, "k_ext_frag");
            NECAT_HIP(ctx, hipEventRecord(ctx->ev[4], s));
            const bool coop = m <= g_coop_threshold;
            const u32 epoch = ++ctx->epoch & 0x3fffffu;
            const bool batch_rc = getenv("NECAT_BATCH_RC") != nullptr;        // the blocks through the checkpoint pass + recomputing walk (ext_rcwalk.h) instead
            if (batch_rc) {
                const size_t per_ck = full ? (size_t)RcGeom<kColsA>::kCk * kWordsA * 16 : (size_t)RcGeom<kColsB>::kCk * kWordsB * 16;
                const size_t per_hc = full ? (size_t)RcGeom<kColsA>::kSeg * kWordsA * 8 : (size_t)RcGeom<kColsB>::kSeg * kWordsB * 8;
                if ((rc2 = buf_ensure(ctx, ctx->scratch[SC_EXT_CKPT], (size_t)g * 64 * (per_ck + per_hc))) ||
                    (rc2 = buf_ensure(ctx, ctx->scratch[SC_EXT_WOUT], (size_t)g * 64 * sizeof(WalkOut)))) return rc2;
                ulonglong2* ck = (ulonglong2*)ctx->scratch[SC_EXT_CKPT].p;
                u64* hcar = (u64*)((char*)ctx->scratch[SC_EXT_CKPT].p + (size_t)g * 64 * per_ck);
                WalkOut* wo = (WalkOut*)ctx->scratch[SC_EXT_WOUT].p;
                const u32 fl = epoch | (1u << 27);
                const bool batch_fast = atoi(getenv("NECAT_BATCH_RC")) == 2;       // .. through the fast general pass k_myers_ckf (both geometries)
                if (full) {
                    if (batch_fast)
                    hipLaunchKernelGGL((k_myers_ckf<kWordsA, kTWordsA, kColsA, 8>), dim3((m + 7) / 8), dim3(64), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag, ck, hcar, error,
                                       d_res, d_stats, epoch | (g_ckr_fast ? 0u : 1u << 28), 0u, g * 64);
                    else
                    hipLaunchKernelGGL((k_myers_ckg<kWordsA, kTWordsA, kColsA, 8>), dim3((m + 7) / 8), dim3(64), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag, ck, hcar, error,
                                       d_res, d_stats, epoch, 0u, g * 64);
                    NECAT_HIP(ctx, hipEventRecord(ctx->ev[5], s));
                    launch_rcwalk2<kWordsA, kTWordsA, kColsA, kOpsA>(m, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag,
                                       (const ulonglong2*)ck, (const u64*)hcar, (const BlockResult*)d_res, (const ExtTask*)nullptr, 1, 1, d_ops, wo, d_stats, d_err, fl, 0u, g * 64);
                    hipLaunchKernelGGL((k_traceback<kWordsA, kTWordsA, kColsA, kOpsA, true, 5>), dim3(g), dim3(64), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag,
                                       (const char*)d_slabs, slab, (const BlockResult*)d_res, d_ops, (ExtTask*)nullptr, 1, d_nops, d_err, ExtLists(), fl, 0u, (const WalkOut*)wo);
                } else {
                    if (batch_fast)
                    hipLaunchKernelGGL((k_myers_ckf<kWordsB, kTWordsB, kColsB, 16>), dim3((m + 3) / 4), dim3(64), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag, ck, hcar, error,
                                       d_res, d_stats, epoch | (g_ckr_fast ? 0u : 1u << 28), 0u, g * 64);
                    else if (atoi(getenv("NECAT_BATCH_RC")) == 64)
                    hipLaunchKernelGGL((k_myers_ckg<kWordsB, kTWordsB, kColsB, 64>), dim3(m), dim3(64), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag, ck, hcar, error,
                                       d_res, d_stats, epoch, 0u, g * 64);
                    else
                    hipLaunchKernelGGL((k_myers_ckg<kWordsB, kTWordsB, kColsB, 16>), dim3((m + 3) / 4), dim3(64), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag, ck, hcar, error,
                                       d_res, d_stats, epoch, 0u, g * 64);
                    NECAT_HIP(ctx, hipEventRecord(ctx->ev[5], s));
                    launch_rcwalk2<kWordsB, kTWordsB, kColsB, kOpsB>(m, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag,
                                       (const ulonglong2*)ck, (const u64*)hcar, (const BlockResult*)d_res, (const ExtTask*)nullptr, 1, 1, d_ops, wo, d_stats, d_err, fl, 0u, g * 64);
                    hipLaunchKernelGGL((k_traceback<kWordsB, kTWordsB, kColsB, kOpsB, true, 5>), dim3(g), dim3(64), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag,
                                       (const char*)d_slabs, slab, (const BlockResult*)d_res, d_ops, (ExtTask*)nullptr, 1, d_nops, d_err, ExtLists(), fl, 0u, (const WalkOut*)wo);
                }
                NECAT_CHECK_LAUNCH(ctx, "k_myers_ckg / k_rcwalk2 / k_traceback");
            } else {
            if (full && coop) {
                const bool f16 = g_fast16 && g_fast >= 1 && g_coop_filter;
                const u32 fl = epoch | (g_coop_filter ? 0u : 1u << 30) | (g_fast == 0 ? 1u << 29 : 0u) | (g_fast == 2 ? 1u << 28 : 0u);
                if (f16) hipLaunchKernelGGL((k_myers_a16<kWordsA, kTWordsA, kColsA>), dim3((m + 15) / 16), dim3(128), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag, d_slabs, slab, error, d_res, d_stats, fl | 1u << 27);
                else hipLaunchKernelGGL((k_myers_coop<kWordsA, kTWordsA, kColsA, 8>), dim3((m + 7) / 8), dim3(64), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag, d_slabs, slab, error, d_res, d_stats, fl, 0u);
            }
            else if (full) hipLaunchKernelGGL((k_myers<kWordsA, kTWordsA, kColsA, true>), dim3(g), dim3(64), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag, d_slabs, slab, error, d_res, d_stats, epoch | ((u32)g_dbg << 28), 0u);
            else if (coop) hipLaunchKernelGGL((k_myers_coop<kWordsB, kTWordsB, kColsB, 16>), dim3((m + 3) / 4), dim3(64), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag, d_slabs, slab, error, d_res, d_stats, epoch | (g_coop_filter ? 0u : 1u << 30) | (g_fast == 0 ? 1u << 29 : 0u) | (g_fast == 2 ? 1u << 28 : 0u), 0u);
            else hipLaunchKernelGGL((k_myers<kWordsB, kTWordsB, kColsB, false>), dim3(g), dim3(64), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, (const u64*)d_frag, d_slabs, slab, error, d_res, d_stats, epoch, 0u);
            NECAT_CHECK_LAUNCH(ctx, "k_myers");
            NECAT_HIP(ctx, hipEventRecord(ctx->ev[5], s));
#define NECAT_TB_LAUNCH(NWX, TWX, COLSX, OPSX, WALK) hipLaunchKernelGGL((k_traceback<NWX, TWX, COLSX, OPSX, true, WALK>), dim3(g), dim3(64), 0, s, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, \
                                         (const u64*)d_frag, (const char*)d_slabs, slab, (const BlockResult*)d_res, d_ops, (ExtTask*)nullptr, 1, d_nops, d_err, ExtLists(), epoch)
            if (full) { if (g_walk == 1) NECAT_TB_LAUNCH(kWordsA, kTWordsA, kColsA, kOpsA, 1); else if (g_walk == 2) NECAT_TB_LAUNCH(kWordsA, kTWordsA, kColsA, kOpsA, 2); else NECAT_TB_LAUNCH(kWordsA, kTWordsA, kColsA, kOpsA, 0); }
            else { if (g_walk == 1) NECAT_TB_LAUNCH(kWordsB, kTWordsB, kColsB, kOpsB, 1); else if (g_walk == 2) NECAT_TB_LAUNCH(kWordsB, kTWordsB, kColsB, kOpsB, 2); else NECAT_TB_LAUNCH(kWordsB, kTWordsB, kColsB, kOpsB, 0); }
#undef NECAT_TB_LAUNCH
            }
            NECAT_CHECK_LAUNCH(ctx, "k_traceback");
            NECAT_HIP(ctx, hipEventRecord(ctx->ev[6], s));
            std::vector<BlockResult> hres(m); std::vector<i32> hn(m); std::vector<u8> hops((size_t)g * 64 * maxops);
            NECAT_HIP(ctx, hipMemcpyAsync(hres.data(), d_res, (size_t)m * sizeof(BlockResult), hipMemcpyDeviceToHost, s));
            NECAT_HIP(ctx, hipMemcpyAsync(hn.data(), d_nops, (size_t)m * 4, hipMemcpyDeviceToHost, s));
            NECAT_HIP(ctx, hipMemcpyAsync(hops.data(), d_ops, hops.size(), hipMemcpyDeviceToHost, s));
            NECAT_HIP(ctx, hipStreamSynchronize(s));
            ctx->tm.myers_ms += ev_ms(ctx->ev[4], ctx->ev[5]); ctx->tm.traceback_ms += ev_ms(ctx->ev[5], ctx->ev[6]); ctx->tm.myers_launches += 1;
            for (u32 j = 0; j < m; ++j) {
                const u64 id = ids[base + j];
                dist[id] = hres[j].dist;
                ctx->tm.myers_word_updates += hres[j].words;
                ctx->tm.myers_cells_bases += (u64)q_len[id] + (u64)t_len[id];
                (void)d_stats;
                if (hres[j].dist >= 0) {
                    const int no = hn[j];
                    std::vector<uint8_t>& f = fwd_ops[id];
                    f.resize((size_t)no);
                    const u8* src = hops.data() + (size_t)(j / 64) * maxops * 64 + (j % 64);
                    int qe = 0, te = 0;
                    for (int x = 0; x < no; ++x) { const u8 op = src[(size_t)(no - 1 - x) * 64]; f[x] = op; qe += op != 2; te += op != 1; }
                    qend[id] = qe; tend[id] = te;
                } else { qend[id] = 0; tend[id] = 0; }
            }
        }
        return NECAT_OK;
    };
    rc = run_shape(itA, idA, true);
    if (!rc) rc = run_shape(itB, idB, false);
    int herr = 0;
    if (!rc) { hipError_t e = hipMemcpy(&herr, d_err, 4, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = set_err(ctx, NECAT_ERR_DEVICE, "memcpy failed"); }
    (void)hipFree(d_err);
    necat_volume_free(ctx, vol);
    if (rc) return rc;
    if (herr) return set_err(ctx, NECAT_ERR_INTERNAL, "edlib kernels reported error code %d", herr);
    if (ops && ops_off) {
        uint64_t* off = (uint64_t*)malloc((n + 1) * 8);
        uint64_t tot = 0;
        for (uint64_t i = 0; i < n; ++i) { off[i] = tot; tot += fwd_ops[i].size(); }
        off[n] = tot;
        uint8_t* o = (uint8_t*)malloc(tot ? tot : 1);
        for (uint64_t i = 0; i < n; ++i) if (!fwd_ops[i].empty()) memcpy(o + off[i], fwd_ops[i].data(), fwd_ops[i].size());
        *ops = o; *ops_off = off;
    }
    return NECAT_OK;
#endif
}
