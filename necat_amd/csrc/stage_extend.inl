// stage_extend.inl - the extension rounds (ext_*.h): lanes, BatchRun, extend_impl, necat_extend / necat_map_pair.
// One of the stage files of libnecat_hip.so's single translation unit: necat_hip.hip includes them in order, inside its extern "C" block, after the
// context / knob / result-pool code they all use (the kernels are header templates and the stages share host helpers: one device code object, one 30 s build).

// ------------------------------------------------------------------------------------------ extension

namespace {

// One batch of candidates advancing through its rounds.  A candidate has one scheduled block at a time; the
// blocks of a round sit in list A (<= 512 x 512) or list B (bigger last blocks).  The chain that bounds the
// run is the list-A chain (frag -> DP -> traceback, round after round), so list B trails it by one round:
//
//     round r    stream a:  A(r)  = blocks of lists[r % 4].A      appends successors to lists[(r + 1) % 4]
//                stream b:  B(r)  = blocks of lists[r % 4].B      appends successors to lists[(r + 2) % 4]
//     lists[r] is complete when A(r - 1) and B(r - 2) are done; B(r) runs under A(r + 1).
//
// The host never waits for the device inside the loop.  The first kernel of A(r) publishes the sizes of lists[r]
// to a pinned ring and resets the counters of lists[(r + 2) % 4]; the host, one round behind, launches B(r - 1)
// with its exact size and A(r) with an upper bound (what was alive a round earlier - every kernel reads the exact
// size on the device); stream-to-stream order is kept by events.  Four list buffers: lists[(r + 2) % 4] receives
// appends from B(r) and A(r + 1) while lists[(r + 1) % 4] is filled by A(r) and B(r - 1), lists[r % 4] is consumed by
// A(r) and B(r), and lists[(r - 1) % 4] may still be read by B(r - 1).  B(r) and B(r - 1) run side by side on two
// streams with two sets of buffers and band pools (list A's pool is reused by A(r + 1) while they run).
struct Batch {
    ExtTask* tasks; u32* count;            // count[4][4]: (full list-A blocks, nB, other list-A blocks, -) per list buffer
    u32 cap;                               // capacity of every item array (list A is filled from both ends)
    BlockItem* itemsA[4]; BlockItem* itemsB[4];
    u64* fragA; u8* opsA; BlockResult* resA;
    // list B: two sets (round parity) - B(r) and B(r - 1) are independent and run side by side
    u64* fragB[2]; u8* opsB[2]; BlockResult* resB[2];
    BlockItem* sortedB[2]; u32* bins[2];    // list B of the round, sorted by size
    hipStream_t sa, sb[2];
    hipEvent_t a0[4], a1[4], a2[4], b0[2], b1[2], b2[2];
    u64 base; u32 n;
};

struct ExtShared {
    const necat_candidate* d_cands; necat_m4* d_m4; u8* d_ok; int* d_err; unsigned long long* stats;
    double error; int tail_match_len, min_align, read_start_id, ref_start_id;
    const u64* reads_off; const u64* ref_off;
    u8* task_ops = nullptr;      // alignment columns per task (necat_onc_align_batch)
};

// One lane of the extension rounds: everything run-to-run state of a batch in flight lives in - arenas, streams, events, its half of the
// published-sizes ring.  Lane 0 is the context's own set; lane 1 (ExtLane1, runtime.h) exists so that the NEXT batch can run its first,
// chip-filling rounds while this one is in its last, latency-bound ones (extend_impl).
enum ExtLaneBuf { LB_TASKS = 0, LB_LISTS, LB_FRAG, LB_OPS, LB_RES, LB_MAT, LB_CKPT, LB_WOUT, LB_CKPTB, LB_CKPTB2, LB_WOUTB, LB_WOUTB2, LB_MATB, LB_MATB2, LB_COUNT };
static_assert(LB_COUNT <= (int)(sizeof(ExtLane1::buf) / sizeof(necat::DevBuf)), "a lane-1 arena without a slot");
struct ExtLane {
    DevBuf *tasks, *lists, *frag, *ops, *res, *mat, *ckpt, *wout, *ckptb[2], *woutb[2], *matb[2];
    hipStream_t sa, sb[2], sd;
    hipEvent_t* ev;                                   // kNumEvents of them, used as necat_ctx::ev is
    volatile RoundPub* ring; RoundPub* ring_dev;      // kRoundRing entries
    unsigned long long* round_seq;
};

int ext_lane(necat_ctx* ctx, int id, ExtLane& L)
{
    if (id == 0) {
        DevBuf* S = ctx->scratch;
        L.tasks = S + SC_EXT_TASKS; L.lists = S + SC_EXT_LISTS; L.frag = S + SC_EXT_FRAG; L.ops = S + SC_EXT_OPS; L.res = S + SC_EXT_RES; L.mat = S + SC_EXT_MAT;
        L.ckpt = S + SC_EXT_CKPT; L.wout = S + SC_EXT_WOUT; L.ckptb[0] = S + SC_EXT_CKPTB; L.ckptb[1] = S + SC_EXT_CKPTB2; L.woutb[0] = S + SC_EXT_WOUTB; L.woutb[1] = S + SC_EXT_WOUTB2;
        L.matb[0] = S + SC_EXT_MATB; L.matb[1] = S + SC_EXT_MATB2;
        L.sa = ctx->stream_a; L.sb[0] = ctx->stream_b; L.sb[1] = ctx->stream_c; L.sd = ctx->stream_d;
        L.ev = ctx->ev;
        L.ring = (volatile RoundPub*)ctx->round_ring; L.ring_dev = (RoundPub*)ctx->round_ring_dev; L.round_seq = &ctx->round_seq;
        return NECAT_OK;
    }
    if (id < 1 || id >= kMaxExtLanes) return set_err(ctx, NECAT_ERR_ARG, "extension lane %d of %d", id, kMaxExtLanes);
    ExtLane1& Q = ctx->lanex[id - 1];
    if (!Q.ready) {
        // Four streams of its own, at the device's LOWEST stream priority (NECAT_LANE1_PRIO: 0 = normal, 1 = lowest - the default -, 2 = highest): the runtime keeps
        // a pool of hardware queues per priority level (GPU_MAX_HW_QUEUES each), so these streams never share a queue with lane 0's - kernels of streams that share
        // a queue run one after the other, and which streams share is the runtime's choice (tools/r05/run22.sh: the same two-lane step took 36 or 45 ms depending on
        // the streams another context had made before) - and lane 0, which holds the longest chains of a call, is served first where both have waves to place.
        static const int lane_prio = getenv("NECAT_LANE1_PRIO") ? atoi(getenv("NECAT_LANE1_PRIO")) : 1;
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
        const int pr = lane_prio == 1 ? least : lane_prio == 2 ? greatest : 0;
        for (hipStream_t& st : Q.st)
            if (!st && (lane_prio && least != greatest ? hipStreamCreateWithPriority(&st, hipStreamDefault, pr) : hipStreamCreate(&st)) != hipSuccess)
                return set_err(ctx, NECAT_ERR_DEVICE, "hipStreamCreate failed (second extension lane)");
        // (events that exist are kept: a creation that failed half-way is completed by the next call, and necat_ctx_destroy destroys every non-null one)
        for (int i = 0; i < kNumEvents; ++i) if (!Q.ev[i] && hipEventCreate(&Q.ev[i]) != hipSuccess) { Q.ev[i] = nullptr; return set_err(ctx, NECAT_ERR_DEVICE, "hipEventCreate failed (second extension lane)"); }
        Q.ready = true;
    }
    DevBuf* S = Q.buf;
    L.tasks = S + LB_TASKS; L.lists = S + LB_LISTS; L.frag = S + LB_FRAG; L.ops = S + LB_OPS; L.res = S + LB_RES; L.mat = S + LB_MAT;
    L.ckpt = S + LB_CKPT; L.wout = S + LB_WOUT; L.ckptb[0] = S + LB_CKPTB; L.ckptb[1] = S + LB_CKPTB2; L.woutb[0] = S + LB_WOUTB; L.woutb[1] = S + LB_WOUTB2;
    L.matb[0] = S + LB_MATB; L.matb[1] = S + LB_MATB2;
    L.sa = Q.st[0]; L.sb[0] = Q.st[1]; L.sb[1] = Q.st[2]; L.sd = Q.st[3];
    L.ev = Q.ev;
    L.ring = (volatile RoundPub*)ctx->round_ring + (size_t)id * kRoundRing; L.ring_dev = (RoundPub*)ctx->round_ring_dev + (size_t)id * kRoundRing; L.round_seq = &Q.round_seq;
    return NECAT_OK;
}

// All rounds of one batch (its first blocks are already in lists[0], appended by k_ext_init on stream a; every list counter but lists[0]'s is
// zero) as a resumable loop: run() is the whole of it; with two lanes (extend_impl) the scheduler calls step() on whichever batch has its next
// sizes published.
struct BatchRun {
    necat_ctx* ctx; const DevVolume& dref; const DevVolume& drd; Batch& c; const ExtShared& X; const ExtLane& L;
    struct Cnt { u32 nA, nB; };
    std::vector<Cnt> hist;                      // published sizes of lists[r]
    std::vector<u32> rc_round;                  // rounds whose full blocks ran through ext_rcwalk.h (L.ev[26 + r % 4] marks the end of the walk kernel)
    std::vector<u8> a_timed;                    // A(r) ran its DP + traceback kernels (events recorded); 2 = as one fused launch (ext_tail.h)
    const unsigned long long seq0;
    volatile RoundPub* const ring;
    RoundPub* const ring_dev;
    bool b_pending[2] = {false, false}, b_fused[2] = {false, false};
    u32 b_blocks[2] = {0, 0};
    double last_wall;
    u32 rnd = 0, launched = 0;                  // the next round to launch; rounds launched
    bool tail = false;                          // fewer than NECAT_EXT_OVERLAP_PCT per cent of the batch's candidates still have a block: the next batch may start beside this one
    bool over = false;                          // nothing alive (or an error): finish() is next
    BatchRun(necat_ctx* ctx_, const DevVolume& dref_, const DevVolume& drd_, Batch& c_, const ExtShared& X_, const ExtLane& L_)
        : ctx(ctx_), dref(dref_), drd(drd_), c(c_), X(X_), L(L_), seq0(*L_.round_seq), ring(L_.ring), ring_dev(L_.ring_dev), last_wall(wall_ms()) {}

    int wait_pub(u32 r, Cnt& out)
    {
        const unsigned long long want = seq0 + r + 1;
        volatile RoundPub* e = &ring[(seq0 + r) % kRoundRing];
        const double t0 = wall_ms();
        for (u64 spin = 0; e->seq != want; ++spin) {
            if ((spin & 0xfffff) == 0xfffff) {
                // a failed kernel never publishes: look at the stream instead of spinning forever
                const hipError_t q = hipStreamQuery(c.sa);
                if (q != hipSuccess && q != hipErrorNotReady) return set_err(ctx, NECAT_ERR_DEVICE, "extension round %u failed: %s", r, hipGetErrorString(q));
                if (wall_ms() - t0 > 120e3) return set_err(ctx, NECAT_ERR_DEVICE, "extension round %u: no progress for 120 s", r);
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        out.nA = e->nA; out.nB = e->nB;
        return NECAT_OK;
    }
    void account_a(u32 r)
    {
        if (r >= a_timed.size() || !a_timed[r]) return;
        const int q = r % 4;
        const u32 nA = hist[r].nA;
        if (a_timed[r] == 2) {
            const double f = ev_ms(c.a0[q], c.a2[q]);
            ctx->tm.fused_ms += f; ctx->tm.fused_launches += 1; ctx->tm.fused_blocks += nA;
            ctx->tm.myers_blocks += nA;
            if (g_trace & 1) fprintf(stderr, "[necat] batch@%lu round %3u: list A %7u blocks  fused DP + walk %.3f ms\n", (unsigned long)c.base, r, nA, f);
            a_timed[r] = 0;
            return;
        }
        const double mA = ev_ms(c.a0[q], c.a1[q]), tA = ev_ms(c.a1[q], c.a2[q]);
        ctx->tm.myers_ms += mA; ctx->tm.traceback_ms += tA;
        if (std::find(rc_round.begin(), rc_round.end(), r) != rc_round.end()) { ctx->tm.rc_ms += ev_ms(c.a1[q], L.ev[26 + (r & 3)]); ctx->tm.rc_ck_ms += mA; ctx->tm.rc_launches += 1; }
        if (nA > g_single_pass) {      // the two-pass instantiation k_myers_coop<8,16,512,8,false> (bench.py's roofline kernel)
            ctx->tm.myersA_ms += mA; ctx->tm.tracebackA_ms += tA; ctx->tm.myersA_launches += 1; ctx->tm.myersA_blocks += nA;
        }
        if (nA > ctx->tm.myersA_big_blocks) { ctx->tm.myersA_big_blocks = nA; ctx->tm.myersA_big_ms = mA; }
        ctx->tm.myers_launches += 1; ctx->tm.myers_blocks += nA;
        if (g_trace & 1) {
            const double now = wall_ms();
            fprintf(stderr, "[necat] batch@%lu round %3u: list A %7u blocks  myers %.3f ms traceback %.3f ms | host wall since last %.3f ms\n",
                    (unsigned long)c.base, r, nA, mA, tA, now - last_wall);
            last_wall = now;
        }
        a_timed[r] = 0;
    }
    void account_b(int slot)
    {
        if (!b_pending[slot]) return;
        if (b_fused[slot]) {
            const double f = ev_ms(c.b0[slot], c.b2[slot]);
            ctx->tm.fused_ms += f; ctx->tm.fused_launches += 1; ctx->tm.fused_blocks += b_blocks[slot]; ctx->tm.myers_blocks += b_blocks[slot];
            if (g_trace & 1) fprintf(stderr, "[necat]          list B: %7u blocks  fused DP + walk %.3f ms\n", b_blocks[slot], f);
            b_pending[slot] = false; b_fused[slot] = false;
            return;
        }
        const double mB = ev_ms(c.b0[slot], c.b1[slot]), tB = ev_ms(c.b1[slot], c.b2[slot]);
        ctx->tm.myers_ms += mB; ctx->tm.traceback_ms += tB;
        ctx->tm.myers_launches += 1; ctx->tm.myers_blocks += b_blocks[slot];
        if (g_trace & 1) fprintf(stderr, "[necat]          list B: %7u blocks  myers %.3f ms traceback %.3f ms\n", b_blocks[slot], mB, tB);
        b_pending[slot] = false;
    }
    // ---- B(q): exact size known (published by A(q)'s first kernel)
    int launch_b(u32 q, u32 nB)
    {
        const int slot = q & 1;
        account_b(slot);                                        // B(q - 2), the previous user of this slot, is done (A(q + 0) started after it)
        const u32 gB = (nB + 63) / 64;
        // small lists (the late rounds, where a round lasts as long as its slowest chain) get alternating streams so
        // that B(q) need not queue behind B(q - 1); big ones stay in one stream - three busy chains only add contention
        hipStream_t sb = c.sb[nB < 4096 ? slot : 0];
        if (g_tail_fused && nB <= g_tail_fused) {
            // a small list: fragments, DP, walk and the next block's plan in one launch, the band in LDS (ext_tail.h)
            const int cur = q % 4, nxt2 = (q + 2) % 4;
            NECAT_HIP(ctx, hipStreamWaitEvent(sb, c.a0[cur], 0));
            NECAT_HIP(ctx, hipStreamWaitEvent(sb, c.b2[slot], 0));
            ExtLists next; next.count = c.count + 4 * nxt2; next.itemsA = c.itemsA[nxt2]; next.itemsB = c.itemsB[nxt2]; next.task_ops = X.task_ops; next.capA = c.cap;
            NECAT_HIP(ctx, hipEventRecord(c.b0[slot], sb));
            hipLaunchKernelGGL((k_tail_fused<kWordsB, kTWordsB, kTailCapB, kOpsB>), dim3(nB), dim3(kTailThreads), 0, sb, drd, dref, (const BlockItem*)c.itemsB[cur], nB,
                               (const u32*)(c.count + 4 * cur + 1), 0u, X.error, c.tasks, X.tail_match_len, X.d_err, next, X.stats);
            NECAT_CHECK_LAUNCH(ctx, "k_tail_fused<B>");
            NECAT_HIP(ctx, hipEventRecord(c.b1[slot], sb));
            NECAT_HIP(ctx, hipEventRecord(c.b2[slot], sb));
            b_pending[slot] = true; b_fused[slot] = true; b_blocks[slot] = nB;
            return NECAT_OK;
        }
        const int cur_b = q % 4, nxt2_b = (q + 2) % 4;
        if (g_rc_listb && g_rc_carry && nB <= g_coop_threshold) {
            // ---- list B through the checkpoint pass + recomputing walk as well (ext_rcwalk.h at 13 words / 16 lanes per block): one DP
            // pass instead of two, no band records, the walk on LDS
            constexpr size_t per_ck = (size_t)RcGeom<kColsB>::kCk * kWordsB * sizeof(ulonglong2), per_hc = (size_t)RcGeom<kColsB>::kSeg * kWordsB * sizeof(u64);
            const u32 rc_chunk = (u32)std::max<size_t>(64, std::min<size_t>((size_t)gB * 64, (g_rc_pool / (per_ck + per_hc)) & ~(size_t)63));
            DevBuf& ckb = *L.ckptb[slot];
            DevBuf& wob = *L.woutb[slot];
            int rc2;
            if ((rc2 = buf_ensure(ctx, ckb, (size_t)rc_chunk * (per_ck + per_hc))) || (rc2 = buf_ensure(ctx, wob, (size_t)gB * 64 * sizeof(WalkOut)))) return rc2;
            ulonglong2* ck = (ulonglong2*)ckb.p;
            u64* hcar = (u64*)((char*)ckb.p + (size_t)rc_chunk * per_ck);
            WalkOut* wo = (WalkOut*)wob.p;
            const BlockItem* itB = c.itemsB[cur_b];
            const u32* d_nB = c.count + 4 * cur_b + 1;
            NECAT_HIP(ctx, hipStreamWaitEvent(sb, c.a0[cur_b], 0));     // lists[q] complete (A(q - 1) done), counters of lists[q + 2] reset
            NECAT_HIP(ctx, hipStreamWaitEvent(sb, c.b2[slot], 0));      // B(q - 2): appended to lists[q], previous user of the slot's buffers
            const u32 epoch = ++ctx->epoch & 0x3fffffu, fl = epoch | (1u << 27);
            ExtLists next; next.count = c.count + 4 * nxt2_b; next.itemsA = c.itemsA[nxt2_b]; next.itemsB = c.itemsB[nxt2_b]; next.task_ops = X.task_ops; next.capA = c.cap;
            RoundCtl ctl; ctl.zero_bins = c.bins[slot];
            hipLaunchKernelGGL((k_ext_frag<kWordsB, kTWordsB>), dim3(grid_for((u64)gB * 64 * kFragSplit, 256)), dim3(256), 0, sb,
                               drd, dref, itB, nB, d_nB, 0u, c.fragB[slot], ctl);
            NECAT_CHECK_LAUNCH(ctx, "k_ext_frag<B>");
            NECAT_HIP(ctx, hipEventRecord(c.b0[slot], sb));
            for (u32 lo = 0; lo < nB; lo += rc_chunk) {
                const u32 hi = std::min<u64>((u64)lo + rc_chunk, (u64)gB * 64), cn = std::min(hi, nB) - lo;
                if (g_rc_fastb)
                hipLaunchKernelGGL((k_myers_ckf<kWordsB, kTWordsB, kColsB, 16>), dim3((cn + 3) / 4), dim3(64), 0, sb, itB, nB, d_nB, 0u, (const u64*)c.fragB[slot], ck, hcar, X.error,
                                   c.resB[slot], X.stats, epoch | (g_ckr_fast ? 0u : 1u << 28), lo, hi);
                else
                hipLaunchKernelGGL((k_myers_ckg<kWordsB, kTWordsB, kColsB, 16>), dim3((cn + 3) / 4), dim3(64), 0, sb, itB, nB, d_nB, 0u, (const u64*)c.fragB[slot], ck, hcar, X.error,
                                   c.resB[slot], X.stats, epoch, lo, hi);
                if (lo + rc_chunk >= nB) NECAT_HIP(ctx, hipEventRecord(c.b1[slot], sb));
                launch_rcwalk2<kWordsB, kTWordsB, kColsB, kOpsB>(cn, sb, itB, nB, d_nB, 0u, (const u64*)c.fragB[slot], (const ulonglong2*)ck,
                                   (const u64*)hcar, (const BlockResult*)c.resB[slot], (const ExtTask*)c.tasks, X.task_ops ? 1 : 0, X.tail_match_len, c.opsB[slot], wo, X.stats, X.d_err, fl, lo, hi);
                NECAT_CHECK_LAUNCH(ctx, "k_myers_ckg / k_rcwalk2<B>");
            }
            hipLaunchKernelGGL((k_traceback<kWordsB, kTWordsB, kColsB, kOpsB, false, 5, kOcaBlockSize, false, 4>), dim3((gB + 3) / 4), dim3(256), 0, sb, itB, nB, d_nB, 0u, (const u64*)c.fragB[slot], (const char*)nullptr, (size_t)0,
                               (const BlockResult*)c.resB[slot], c.opsB[slot], c.tasks, X.tail_match_len, (i32*)nullptr, X.d_err, next, fl, 0u, (const WalkOut*)wo);
            NECAT_CHECK_LAUNCH(ctx, "k_traceback<B, rc>");
            NECAT_HIP(ctx, hipEventRecord(c.b2[slot], sb));
            b_pending[slot] = true; b_blocks[slot] = nB;
            return NECAT_OK;
        }
#if !NECAT_XCHECK
        NECAT_RETIRED(ctx, "list B through the band-record kernels (NECAT_RC_LISTB=0 / NECAT_RC_CARRY=0 / NECAT_COOP_THRESHOLD)");
#else
        DevBuf& poolB = *L.matb[slot];
        // a capped band pool (NECAT_BAND_POOL_MB): the list in chunks of what the pool holds, DP + walk per chunk
        u32 gchunk = gB;
        if (g_band_pool && (size_t)gB * kSlabB > g_band_pool) gchunk = (u32)std::max<size_t>(1, g_band_pool / kSlabB);
        if ((size_t)gchunk * kSlabB > poolB.cap) {
            const size_t need = (size_t)gchunk * kSlabB;
            int rc = ensure_zeroed(ctx, poolB, gchunk < gB ? need : need + need / 4, sb);
            if (rc) return rc;
        }
        const int cur = q % 4, nxt2 = (q + 2) % 4;
        const BlockItem* itB = c.itemsB[cur];
        const u32* d_nB = c.count + 4 * cur + 1;
        NECAT_HIP(ctx, hipStreamWaitEvent(sb, c.a0[cur], 0));     // lists[q] complete (A(q - 1) done), counters of lists[q + 2] reset
        NECAT_HIP(ctx, hipStreamWaitEvent(sb, c.b2[slot], 0));    // B(q - 2): appended to lists[q], previous user of the slot's buffers
        // (B(q - 1) on the other stream reads lists[q - 1] and appends to lists[q + 1]; this round appends to lists[q + 2]:
        // four list buffers keep the two apart - with three, lists[q + 2] WAS lists[q - 1])
        const u32 epoch = ++ctx->epoch & 0x3fffffu;
        ExtLists next; next.count = c.count + 4 * nxt2; next.itemsA = c.itemsA[nxt2]; next.itemsB = c.itemsB[nxt2]; next.task_ops = X.task_ops; next.capA = c.cap;
        // (below ~2 k blocks every wave is resident at once and the round lasts as long as its longest walk: order is irrelevant)
        if (nB >= 2048 && g_sort_b) {
            hipLaunchKernelGGL(k_items_hist, dim3(grid_for(nB, 256)), dim3(256), 0, sb, itB, nB, c.bins[slot]);
            hipLaunchKernelGGL(k_items_scan, dim3(1), dim3(64), 0, sb, c.bins[slot]);
            hipLaunchKernelGGL(k_items_scatter, dim3(grid_for(nB, 256)), dim3(256), 0, sb, itB, nB, c.bins[slot], c.sortedB[slot]);
            NECAT_CHECK_LAUNCH(ctx, "k_items_sort");
            itB = c.sortedB[slot];
        }
        RoundCtl ctl; ctl.zero_bins = c.bins[slot];
        hipLaunchKernelGGL((k_ext_frag<kWordsB, kTWordsB>), dim3(grid_for((u64)gB * 64 * kFragSplit, 256)), dim3(256), 0, sb,
                           drd, dref, itB, nB, d_nB, 0u, c.fragB[slot], ctl);
        NECAT_CHECK_LAUNCH(ctx, "k_ext_frag<B>");
        NECAT_HIP(ctx, hipEventRecord(c.b0[slot], sb));
        for (u32 g0 = 0; g0 < gB; g0 += gchunk) {
            const u32 lo = g0 * 64, hi = std::min(nB, (g0 + gchunk) * 64), cn = hi - lo;       // work items of this chunk
            char* slabsB = (char*)poolB.p - (size_t)g0 * kSlabB;                              // the kernels index slabs by item / 64
            if (nB <= g_single_pass && nB <= g_coop_threshold)
                hipLaunchKernelGGL((k_myers_coop<kWordsB, kTWordsB, kColsB, 16, true>), dim3((cn + 3) / 4), dim3(64), 0, sb, itB, hi, d_nB, 0u,
                                   (const u64*)c.fragB[slot], slabsB, kSlabB, X.error, c.resB[slot], X.stats, epoch, lo);
            else if (nB <= g_coop_threshold)
                hipLaunchKernelGGL((k_myers_coop<kWordsB, kTWordsB, kColsB, 16>), dim3((cn + 3) / 4), dim3(64), 0, sb, itB, hi, d_nB, 0u,
                                   (const u64*)c.fragB[slot], slabsB, kSlabB, X.error, c.resB[slot], X.stats, epoch | (g_coop_filter ? 0u : 1u << 30) | (g_fast == 0 ? 1u << 29 : 0u) | (g_fast == 2 ? 1u << 28 : 0u), lo);
            else
                hipLaunchKernelGGL((k_myers<kWordsB, kTWordsB, kColsB, false>), dim3((cn + 63) / 64), dim3(64), 0, sb, itB, hi, d_nB, 0u,
                                   (const u64*)c.fragB[slot], slabsB, kSlabB, X.error, c.resB[slot], X.stats, epoch, lo);
            NECAT_CHECK_LAUNCH(ctx, "k_myers<B>");
            if (g0 + gchunk >= gB) NECAT_HIP(ctx, hipEventRecord(c.b1[slot], sb));
#define NECAT_TB_LAUNCH(WALK) hipLaunchKernelGGL((k_traceback<kWordsB, kTWordsB, kColsB, kOpsB, false, WALK>), dim3((cn + 63) / 64), dim3(64), 0, sb, itB, hi, d_nB, 0u, \
                           (const u64*)c.fragB[slot], (const char*)slabsB, kSlabB, (const BlockResult*)c.resB[slot], c.opsB[slot], c.tasks, X.tail_match_len, \
                           (i32*)nullptr, X.d_err, next, epoch, lo)
            if (g_walk_wave && nB <= g_walk_wave)       // a small list: one wave per block, band records through an LDS window
                hipLaunchKernelGGL((k_walk_wave<kWordsB, kTWordsB, kOpsB>), dim3(cn), dim3(64), 0, sb, itB, hi, d_nB, 0u, (const u64*)c.fragB[slot], (const char*)slabsB, kSlabB,
                                   (const BlockResult*)c.resB[slot], c.tasks, X.tail_match_len, X.d_err, next, lo);
            else if (g_walk == 1) NECAT_TB_LAUNCH(1); else if (g_walk == 2) NECAT_TB_LAUNCH(2); else if (g_walk == 3) NECAT_TB_LAUNCH(3); else if (g_walk == 4) NECAT_TB_LAUNCH(4); else NECAT_TB_LAUNCH(0);
#undef NECAT_TB_LAUNCH
            NECAT_CHECK_LAUNCH(ctx, "k_traceback<B>");
        }
        NECAT_HIP(ctx, hipEventRecord(c.b2[slot], sb));
        b_pending[slot] = true; b_blocks[slot] = nB;
        return NECAT_OK;
#endif
    }
    // ---- A(r): grid sized by an upper bound, the kernels read the exact size of lists[r]
    int launch_a(u32 r, u32 bound)
    {
        const int cur = r % 4, nxt = (r + 1) % 4, nxt2 = (r + 2) % 4;
        if (g_tail_fused && bound && bound <= g_tail_fused) {
            // a small list: one launch for the round (ext_tail.h); the round's bookkeeping first, as a launch of its own - list B's
            // chain of this round waits for a0, not for the fused kernel
            if (r >= 2) NECAT_HIP(ctx, hipStreamWaitEvent(c.sa, c.b2[r & 1], 0));        // B(r - 2) appended to lists[r]
            const u32* d_nA = c.count + 4 * cur;
            RoundCtl ctl; ctl.count = d_nA; ctl.zero = c.count + 4 * nxt2; ctl.seq = seq0 + r + 1; ctl.pub = ring_dev + (seq0 + r) % kRoundRing;
            hipLaunchKernelGGL(k_round_ctl, dim3(1), dim3(64), 0, c.sa, ctl);
            NECAT_CHECK_LAUNCH(ctx, "k_round_ctl");
            NECAT_HIP(ctx, hipEventRecord(c.a0[cur], c.sa));
            ExtLists next; next.count = c.count + 4 * nxt; next.itemsA = c.itemsA[nxt]; next.itemsB = c.itemsB[nxt]; next.task_ops = X.task_ops; next.capA = c.cap;
            hipLaunchKernelGGL((k_tail_fused<kWordsA, kTWordsA, kColsA * kWordsA, kOpsA>), dim3(bound), dim3(kTailThreads), 0, c.sa, drd, dref, (const BlockItem*)c.itemsA[cur], bound,
                               d_nA, c.cap, X.error, c.tasks, X.tail_match_len, X.d_err, next, X.stats);
            NECAT_CHECK_LAUNCH(ctx, "k_tail_fused<A>");
            NECAT_HIP(ctx, hipEventRecord(c.a1[cur], c.sa));
            NECAT_HIP(ctx, hipEventRecord(c.a2[cur], c.sa));
            a_timed.push_back(2);
            return NECAT_OK;
        }
        const u32 gA = (bound + 63) / 64;
        // the band pools are sized by what a round needs (round 0 of the first call sets them: 35 GB instead of the
        // 76 GB worst case "every block in list B" at E. coli size - hipMalloc costs ~13 ms per GB); with a capped pool
        // (NECAT_BAND_POOL_MB, the command-line programs: a fresh process pays 30 - 55 ms per GB of VRAM the previous one
        // dirtied) the list runs in chunks of what the pool holds, DP + walk per chunk
        u32 gchunk = gA;
        if (g_band_pool && (size_t)gA * kSlabA > g_band_pool) gchunk = (u32)std::max<size_t>(1, g_band_pool / kSlabA);
        // a big round through ext_rcwalk.h (checkpoints + recomputing walk): no band records at all when its ragged blocks go the same way
        const bool wide_possible = g_rc_maxdist < (int)((double)kOcaBlockSize * X.error * 1.1);       // (edlib_ex.c:751: no block has a larger distance)
        const bool rc_band = !g_rc_ragged || wide_possible;                                            // the round still needs the band pool (whole list: slabs are indexed by work index)
        const bool use_rc = g_rcwalk && bound > g_rcwalk && bound <= g_coop_threshold && g_fast == 1 && g_coop_filter && (!rc_band || gchunk == gA);
        if ((!use_rc || rc_band) && (size_t)gchunk * kSlabA > (*L.mat).cap) {
            const size_t need = (size_t)gchunk * kSlabA;
            int rc = ensure_zeroed(ctx, (*L.mat), gchunk < gA ? need : need + need / 8, c.sa);
            if (rc) return rc;
        }
#if !NECAT_XCHECK
        if (!use_rc || !g_rc_carry || !g_rc_ragged || wide_possible)
            NECAT_RETIRED(ctx, "list A through the band-record kernels (NECAT_RCWALK=0, NECAT_TAIL_FUSED=0 without NECAT_RCWALK=1, NECAT_RC_CARRY=0, NECAT_RC_RAGGED=0, NECAT_RC_MAXDIST, NECAT_FAST, NECAT_COOP_*)");
#endif
        const BlockItem* itA = c.itemsA[cur];
        const u32* d_nA = c.count + 4 * cur;            // [0] full blocks (front of itemsA), [2] the others (back)
        if (r >= 2) NECAT_HIP(ctx, hipStreamWaitEvent(c.sa, c.b2[r & 1], 0));        // B(r - 2) appended to lists[r]
        const u32 epoch = ++ctx->epoch & 0x3fffffu;
        RoundCtl ctl; ctl.count = d_nA; ctl.zero = c.count + 4 * nxt2; ctl.seq = seq0 + r + 1; ctl.pub = ring_dev + (seq0 + r) % kRoundRing;
        // NECAT_FRAG_FUSE (round 6): the merged big-round path cuts its fragments inside the checkpoint pass (k_myers_ck flag bit 22); the round's bookkeeping - list sizes
        // published, the counters of the list after next reset - is then the one-wave k_round_ctl, and list B's chain of the round (which waits for a0) starts that much earlier
        const bool fuse_frag = g_frag_fuse && use_rc && g_rc_carry && g_rc_ragged && g_rc_merge && !wide_possible && !getenv("NECAT_RC_CKG_ALL") && bound > 0;
        if (fuse_frag) {
            hipLaunchKernelGGL(k_round_ctl, dim3(1), dim3(64), 0, c.sa, ctl);
            NECAT_CHECK_LAUNCH(ctx, "k_round_ctl");
        } else {
        hipLaunchKernelGGL((k_ext_frag<kWordsA, kTWordsA>), dim3(grid_for((u64)std::max(gA, 1u) * 64 * kFragSplit, 256)), dim3(256), 0, c.sa,
                           drd, dref, itA, bound, d_nA, c.cap, c.fragA, ctl);
        NECAT_CHECK_LAUNCH(ctx, "k_ext_frag<A>");
        }
        NECAT_HIP(ctx, hipEventRecord(c.a0[cur], c.sa));
        a_timed.push_back(0);
        if (!bound) return NECAT_OK;
        ExtLists next; next.count = c.count + 4 * nxt; next.itemsA = c.itemsA[nxt]; next.itemsB = c.itemsB[nxt]; next.task_ops = X.task_ops; next.capA = c.cap;
        if (use_rc) {
            // ---- a big round: the full blocks (the front of the work index space) without NW pass and band records - SHW with
            // checkpoints, then the walk that recomputes its cells (ext_rcwalk.h); the ragged blocks and the few blocks whose band is
            // too wide for that walk through the usual kernels, in the same launches (epoch bit 24)
            int rc2;
            // checkpoints (+ deltas) of at most g_rc_pool bytes: a longer list goes through the buffer in several launches, one after the other on stream a
            const size_t per_item = (size_t)(g_rc_carry ? kRcCk16 : kRcCk) * 8 * sizeof(ulonglong2), per_item_hc = g_rc_carry ? (size_t)kRcCk * 8 * sizeof(u64) : 0;
            const u32 rc_chunk = (u32)std::max<size_t>(64, std::min<size_t>((size_t)gA * 64, (g_rc_pool / (per_item + per_item_hc)) & ~(size_t)63));
            const size_t ck_bytes = (size_t)rc_chunk * per_item;
            if ((rc2 = buf_ensure(ctx, (*L.ckpt), ck_bytes + (size_t)rc_chunk * per_item_hc)) ||
                (rc2 = buf_ensure(ctx, (*L.wout), (size_t)gA * 64 * sizeof(WalkOut)))) return rc2;
            ulonglong2* ck = (ulonglong2*)(*L.ckpt).p;
            u64* hcar = (u64*)((char*)(*L.ckpt).p + ck_bytes);
            WalkOut* wo = (WalkOut*)(*L.wout).p;
            char* slabsA = (char*)(*L.mat).p;
            // the ragged blocks (and, once k_myers_ck has flagged them, the wide ones) on a stream of their own: a lane-per-block walk
            // of a tenth of the list is as long as one of the whole list (latency bound) - it runs beside the full blocks' chain
            hipStream_t sd = L.sd;
            const u32 fl_rag = epoch | (1u << 26), fl_wide = epoch | (1u << 25), fl_all = g_rc_ragged ? epoch | (1u << 27) : epoch;
#if NECAT_XCHECK
            if (!g_rc_ragged) {
                NECAT_HIP(ctx, hipStreamWaitEvent(sd, c.a0[cur], 0));            // the fragments are there
                hipLaunchKernelGGL((k_myers_coop<kWordsA, kTWordsA, kColsA, 8>), dim3(gA * 8), dim3(64), 0, sd, itA, bound, d_nA, c.cap,
                                   (const u64*)c.fragA, slabsA, kSlabA, X.error, c.resA, X.stats, fl_rag, 0u);
                hipLaunchKernelGGL((k_traceback<kWordsA, kTWordsA, kColsA, kOpsA, false, 0>), dim3(gA), dim3(64), 0, sd, itA, bound, d_nA, c.cap,
                                   (const u64*)c.fragA, (const char*)slabsA, kSlabA, (const BlockResult*)c.resA, c.opsA, c.tasks, X.tail_match_len,
                                   (i32*)nullptr, X.d_err, next, fl_rag, 0u);
                NECAT_CHECK_LAUNCH(ctx, "k_myers / k_traceback<A, ragged>");
            }
#endif
            // the full blocks on stream a: SHW + checkpoints, recompute walk (chunk by chunk), finish
            const bool one_chunk = rc_chunk >= bound;
            static const bool ckg_all = getenv("NECAT_RC_CKG_ALL") != nullptr;       // debugging: every block through the general pass
            // NECAT_RC_MERGE (default): the ragged blocks ride the same two launches as the full ones (k_myers_ck's ragged fast path, the walk
            // over the whole list) instead of a chain of their own (k_myers_ckg + walk on stream d)
            const bool merged = g_rc_merge && g_rc_ragged && g_rc_carry && !ckg_all;
            // NECAT_RC_PIPE (default 1 = off): a big list in that many pieces, the walk of piece i on stream d beside the checkpoint pass of piece
            // i + 1 on stream a - the pass is bound by VALU issue, the walk by the latency of its one walker wave per 64 blocks (a third of the
            // pass's instruction rate), and one after the other they are the critical chain of every big round.  Measured: both kernels just
            // take longer side by side, 41.6 -> 43.4 - 43.9 ms per step with 2 - 4 pieces, with or without raised priority for the walk
            const bool piped = g_rc_pipe > 1 && one_chunk && merged && !wide_possible && bound >= g_rc_pipe_min;
            const u32 step_chunk = piped ? (u32)(((((u64)gA * 64 + g_rc_pipe - 1) / g_rc_pipe) + 63) & ~63ULL) : rc_chunk;
            int ci = 0;
            for (u32 lo = 0; lo < bound; lo += step_chunk, ++ci) {
                const u32 hi = std::min<u64>((u64)lo + step_chunk, (u64)gA * 64), cn = hi - lo;
                const bool last = (u64)lo + step_chunk >= bound;
                // (a piece's checkpoints and deltas at its own place in the buffer, which holds the whole list then: the kernels index by item - lo)
                ulonglong2* const ck_all = ck; u64* const hcar_all = hcar;
                ulonglong2* const ck = piped ? ck_all + (size_t)lo * (per_item / sizeof(ulonglong2)) : ck_all;
                u64* const hcar = piped ? hcar_all + (size_t)lo * (per_item_hc / sizeof(u64)) : hcar_all;
                hipStream_t sw = piped ? sd : c.sa;
                if (ckg_all && g_rc_ragged) {}
                else if (g_rc_carry)
                    hipLaunchKernelGGL((k_myers_ck<kWordsA, kTWordsA, true>), dim3((cn + 7) / 8), dim3(64), g_ck_lds, c.sa, itA, d_nA, c.cap, (const u64*)c.fragA, ck, hcar, X.error, c.resA, X.stats, g_rc_maxdist, lo, hi,
                                       (merged ? fl_all : epoch) | (g_ck_post ? 0u : 1u << 24) | (g_rc_prio & 2u ? 1u << 23 : 0u) | (fuse_frag && merged ? 1u << 22 : 0u),
                                       (const u64*)drd.bases, (const u64*)dref.bases);
                else
#if NECAT_XCHECK
                    hipLaunchKernelGGL((k_myers_ck<kWordsA, kTWordsA, false>), dim3((cn + 7) / 8), dim3(64), 0, c.sa, itA, d_nA, c.cap, (const u64*)c.fragA, ck, hcar, X.error, c.resA, X.stats, g_rc_maxdist, lo, hi, epoch);
#else
                    {}
#endif
                if (piped) { NECAT_HIP(ctx, hipEventRecord(L.ev[40 + (ci & 7)], c.sa)); NECAT_HIP(ctx, hipStreamWaitEvent(sw, L.ev[40 + (ci & 7)], 0)); }
                if (g_rc_ragged && !merged) {
                    // the ragged blocks of the chunk (the back of the work index space): the general SHW pass, same checkpoints.  A tenth of
                    // the blocks, few waves, latency bound: beside the full blocks' pass on a stream of its own when the list is one chunk
                    hipStream_t sr = one_chunk ? sd : c.sa;
                    if (one_chunk) NECAT_HIP(ctx, hipStreamWaitEvent(sd, c.a0[cur], 0));            // the fragments are there
                    hipLaunchKernelGGL((k_myers_ckg<kWordsA, kTWordsA, kColsA, 8>), dim3((cn + 7) / 8), dim3(64), 0, sr, itA, bound, d_nA, c.cap, (const u64*)c.fragA, ck, hcar, X.error,
                                       c.resA, X.stats, ckg_all ? epoch : fl_rag, lo, hi);
                    if (one_chunk) {       // .. and their walk there too: the full blocks' walk need not wait for this pass (as long as the full blocks' own)
                        launch_rcwalk2<kWordsA, kTWordsA, kColsA, kOpsA>(cn, sd, itA, bound, d_nA, c.cap, (const u64*)c.fragA, (const ulonglong2*)ck,
                                           (const u64*)hcar, (const BlockResult*)c.resA, (const ExtTask*)c.tasks, X.task_ops ? 1 : 0, X.tail_match_len, c.opsA, wo, X.stats, X.d_err, fl_rag, lo, hi);
                        NECAT_HIP(ctx, hipEventRecord(L.ev[30], sd));
                    }
                }
                NECAT_CHECK_LAUNCH(ctx, "k_myers_ck");
                if (last) NECAT_HIP(ctx, hipEventRecord(c.a1[cur], c.sa));
                if (g_rc_carry)
                    launch_rcwalk2<kWordsA, kTWordsA, kColsA, kOpsA>(cn, sw, itA, bound, d_nA, c.cap, (const u64*)c.fragA, (const ulonglong2*)ck,
                                       (const u64*)hcar, (const BlockResult*)c.resA, (const ExtTask*)c.tasks, X.task_ops ? 1 : 0, X.tail_match_len, c.opsA, wo, X.stats, X.d_err,
                                       (g_rc_ragged && one_chunk && !merged) ? epoch : fl_all, lo, hi);
                else
#if NECAT_XCHECK
                    hipLaunchKernelGGL((k_rcwalk4<kWordsA, kTWordsA, kOpsA>), dim3((cn + 15) / 16), dim3(64), 0, c.sa, itA, d_nA, c.cap, (const u64*)c.fragA, (const ulonglong2*)ck,
                                       (const BlockResult*)c.resA, (const ExtTask*)c.tasks, X.task_ops ? 1 : 0, X.tail_match_len, c.opsA, wo, X.stats, X.d_err, lo, hi);
#else
                    {}
#endif
                NECAT_CHECK_LAUNCH(ctx, "k_rcwalk");
            }
            NECAT_HIP(ctx, hipEventRecord(L.ev[26 + (r & 3)], piped ? sd : c.sa));       // a1 -> this: the walk kernel alone (account_a; of the last chunk, normally the only one)
            if (piped) NECAT_HIP(ctx, hipStreamWaitEvent(c.sa, L.ev[26 + (r & 3)], 0));          // the finishing kernel reads what the walks left
#if NECAT_XCHECK
            if (wide_possible) {
                NECAT_HIP(ctx, hipStreamWaitEvent(sd, c.a1[cur], 0));            // k_myers_ck has flagged the wide blocks
                hipLaunchKernelGGL((k_myers_coop<kWordsA, kTWordsA, kColsA, 8>), dim3(gA * 8), dim3(64), 0, sd, itA, bound, d_nA, c.cap,
                                   (const u64*)c.fragA, slabsA, kSlabA, X.error, c.resA, X.stats, fl_wide, 0u);
                hipLaunchKernelGGL((k_traceback<kWordsA, kTWordsA, kColsA, kOpsA, false, 0>), dim3(gA), dim3(64), 0, sd, itA, bound, d_nA, c.cap,
                                   (const u64*)c.fragA, (const char*)slabsA, kSlabA, (const BlockResult*)c.resA, c.opsA, c.tasks, X.tail_match_len,
                                   (i32*)nullptr, X.d_err, next, fl_wide, 0u);
                NECAT_CHECK_LAUNCH(ctx, "k_myers / k_traceback<A, wide>");
            }
#endif
            if (!g_rc_ragged || wide_possible) NECAT_HIP(ctx, hipEventRecord(L.ev[25], sd));
            if (g_rc_ragged && one_chunk && !(g_rc_merge && g_rc_carry && !getenv("NECAT_RC_CKG_ALL"))) NECAT_HIP(ctx, hipStreamWaitEvent(c.sa, L.ev[30], 0));       // the ragged blocks are walked
            rc_round.push_back(r);
            hipLaunchKernelGGL((k_traceback<kWordsA, kTWordsA, kColsA, kOpsA, false, 5, kOcaBlockSize, false, 4>), dim3((gA + 3) / 4), dim3(256), 0, c.sa, itA, bound, d_nA, c.cap,
                               (const u64*)c.fragA, (const char*)slabsA, kSlabA, (const BlockResult*)c.resA, c.opsA, c.tasks, X.tail_match_len,
                               (i32*)nullptr, X.d_err, next, fl_all, 0u, (const WalkOut*)wo);
            NECAT_CHECK_LAUNCH(ctx, "k_traceback<A, rc>");
            if (!g_rc_ragged || wide_possible) NECAT_HIP(ctx, hipStreamWaitEvent(c.sa, L.ev[25], 0));          // the round is over when both chains are
        }
#if NECAT_XCHECK
        else
        for (u32 g0 = 0; g0 < gA; g0 += gchunk) {
            const u32 lo = g0 * 64, hi = std::min(gA, g0 + gchunk) * 64, cn = hi - lo;           // work indices of this chunk (the kernels know the exact list)
            char* slabsA = (char*)(*L.mat).p - (size_t)g0 * kSlabA;             // the kernels index slabs by work index / 64
            if (bound <= g_single_pass && bound <= g_coop_threshold)
                hipLaunchKernelGGL((k_myers_coop<kWordsA, kTWordsA, kColsA, 8, true>), dim3(cn / 8), dim3(64), 0, c.sa, itA, bound, d_nA, c.cap,
                                   (const u64*)c.fragA, slabsA, kSlabA, X.error, c.resA, X.stats, epoch, lo);
            else if (bound <= g_coop_threshold) {
                const bool f16 = g_fast16 && g_fast == 1 && g_coop_filter && gchunk == gA;
                const u32 fl = epoch | (g_coop_filter ? 0u : 1u << 30) | (g_fast == 0 ? 1u << 29 : 0u) | (g_fast == 2 ? 1u << 28 : 0u);
                if (f16)      // workgroups of 16 work items: 16 full blocks take the 16-block path (ext_fast16.h), anything else the general one
                    hipLaunchKernelGGL((k_myers_a16<kWordsA, kTWordsA, kColsA>), dim3((bound + 15) / 16), dim3(128), 0, c.sa, itA, bound, d_nA, c.cap,
                                       (const u64*)c.fragA, slabsA, kSlabA, X.error, c.resA, X.stats, fl | 1u << 27);
                else
                    hipLaunchKernelGGL((k_myers_coop<kWordsA, kTWordsA, kColsA, 8>), dim3(cn / 8), dim3(64), 0, c.sa, itA, bound, d_nA, c.cap,
                                       (const u64*)c.fragA, slabsA, kSlabA, X.error, c.resA, X.stats, fl, lo);
            }
            else
                hipLaunchKernelGGL((k_myers<kWordsA, kTWordsA, kColsA, false>), dim3(cn / 64), dim3(64), 0, c.sa, itA, bound, d_nA, c.cap,   // list A also holds last blocks <= 512 x 512
                                   (const u64*)c.fragA, slabsA, kSlabA, X.error, c.resA, X.stats, epoch, lo);
            NECAT_CHECK_LAUNCH(ctx, "k_myers<A>");
            if (g0 + gchunk >= gA) NECAT_HIP(ctx, hipEventRecord(c.a1[cur], c.sa));
#define NECAT_TB_LAUNCH(WALK) hipLaunchKernelGGL((k_traceback<kWordsA, kTWordsA, kColsA, kOpsA, false, WALK>), dim3(cn / 64), dim3(64), 0, c.sa, itA, bound, d_nA, c.cap, \
                           (const u64*)c.fragA, (const char*)slabsA, kSlabA, (const BlockResult*)c.resA, c.opsA, c.tasks, X.tail_match_len, \
                           (i32*)nullptr, X.d_err, next, epoch, lo)
            if (g_walk_wave && bound <= g_walk_wave)
                hipLaunchKernelGGL((k_walk_wave<kWordsA, kTWordsA, kOpsA>), dim3(cn), dim3(64), 0, c.sa, itA, bound, d_nA, c.cap, (const u64*)c.fragA, (const char*)slabsA, kSlabA,
                                   (const BlockResult*)c.resA, c.tasks, X.tail_match_len, X.d_err, next, lo);
            else if (g_walk == 1) NECAT_TB_LAUNCH(1); else if (g_walk == 2) NECAT_TB_LAUNCH(2); else if (g_walk == 3) NECAT_TB_LAUNCH(3); else if (g_walk == 4) NECAT_TB_LAUNCH(4); else NECAT_TB_LAUNCH(0);
#undef NECAT_TB_LAUNCH
            NECAT_CHECK_LAUNCH(ctx, "k_traceback<A>");
        }
#endif
        NECAT_HIP(ctx, hipEventRecord(c.a2[cur], c.sa));
        a_timed[r] = 1;
        return NECAT_OK;
    }
    // the sizes of lists[rnd] have been published (round 0: k_ext_init filled them): step() will not wait
    bool ready() const { return rnd == 0 || ring[(seq0 + rnd - 1) % kRoundRing].seq == seq0 + rnd; }
    // one turn of the round loop: the published sizes of lists[rnd], list B of round rnd - 1, list A of round rnd
    int step()
    {
        int rc;
        u32 bound = c.n + 16;
        if (rnd > 0) {
            Cnt prev;
            if ((rc = wait_pub(rnd - 1, prev))) { over = true; return rc; }          // A(rnd - 1) has started: A(rnd - 2) and B(rnd - 3) are done
            hist.push_back(prev);
            if (rnd >= 2) account_a(rnd - 2);
            const u32 nB2 = rnd >= 2 ? hist[rnd - 2].nB : 0;     // B(rnd - 2) may still be running: its successors join lists[rnd]
            const u64 alive = (u64)prev.nA + prev.nB + nB2;
            if (alive * 100 < (u64)c.n * g_ext_overlap_pct) tail = true;
            if (alive == 0) { over = tail = true; return NECAT_OK; }          // nothing alive
            if (prev.nB) { if ((rc = launch_b(rnd - 1, prev.nB))) { over = true; return rc; } }
            bound = prev.nA + nB2 + 16;                     // work indices: the full blocks rounded up to 16, then the others
        }
        if ((rc = launch_a(rnd, bound))) { over = true; return rc; }
        launched = ++rnd;
        return NECAT_OK;
    }
    // nothing of this batch is in flight any more (two lanes: the scheduler polls this instead of blocking in finish())
    bool drained() const
    {
        for (hipStream_t s : {c.sa, c.sb[0], c.sb[1]}) if (hipStreamQuery(s) == hipErrorNotReady) { (void)hipGetLastError(); return false; }      // ("not ready" is no error to the next launch check)
        return true;
    }
    // drain whatever is still in flight, the last rounds' accounts; rc = what step() returned
    int finish(int rc)
    {
        over = tail = true;
        hipError_t e1 = hipStreamSynchronize(c.sa), e2 = hipStreamSynchronize(c.sb[0]), e3 = hipStreamSynchronize(c.sb[1]);
        *L.round_seq = seq0 + launched;
        if (!rc) for (hipError_t e : {e1, e2, e3}) if (e != hipSuccess) rc = set_err(ctx, NECAT_ERR_DEVICE, "extension rounds: %s", hipGetErrorString(e));
        if (rc) return rc;
        if (launched) {
            // the last launched round published too (its lists are empty unless the loop ended on an error)
            Cnt last; if ((rc = wait_pub(launched - 1, last))) return rc;
            if (hist.size() < launched) hist.push_back(last);
            if (launched >= 2) account_a(launched - 2);
            account_a(launched - 1);
        }
        account_b(0); account_b(1);
        for (const Cnt& h : hist) ctx->tm.rounds += (h.nA + h.nB) ? 1 : 0;
        return NECAT_OK;
    }
    // all rounds, one after the other (one lane)
    int run()
    {
        int rc = NECAT_OK;
        while (!over && !(rc = step())) {}
        return finish(rc);
    }
};

}  // namespace

namespace {
// outputs of the alignment-keeping mode (necat_onc_align_batch)
struct AlignOut {
    necat_alignment* aln = nullptr;
    std::vector<std::pair<u8*, u64>> parts;     // one pinned block of columns per batch
    u64 total = 0;
    std::vector<u64> off;
    bool defer_copy = false;    // the columns' device-to-host copy runs on ctx->stream_copy and is NOT waited for: the caller
                                // synchronises that stream before it reads (or frees) the blocks
};

// The extension loop behind necat_extend (M4 records, containment filter) and necat_onc_align_batch
// (every candidate's alignment with its columns, `ao` != nullptr).
struct DevOut { const necat_m4* d = nullptr; uint64_t n = 0; };      // records left on the device (sharded calls gather them there)
// read-to-reference mapping (necat_map_reference): every candidate aligned against its stretch of the reference (rm_window), and
// instead of the filtered records every candidate's own record + flag come back, with the candidates: the caller's loop decides
struct RmOut { std::vector<necat_candidate> cands; std::vector<necat_m4> m4; std::vector<u8> ok; std::vector<u64> group_off; };

int ext_streams(necat_ctx* ctx, bool with_copy = false)
{
    // NECAT_SERIAL=1 (profiling): the four streams of the extension rounds are ONE stream, so that every kernel has the chip to itself and its
    // duration is its own work, not its wait for wave slots behind the other chains (tools/r04_profile.sh: the exclusive-time table)
    static const bool serial = getenv("NECAT_SERIAL") && atoi(getenv("NECAT_SERIAL"));
    if (serial && !ctx->stream_a) { ctx->stream_a = ctx->stream_b = ctx->stream_c = ctx->stream_d = ctx->stream; ctx->serial_streams = true; }
    // NECAT_STREAM_PRIO=1: the streams of list B and of the ragged / wide blocks at the device's highest priority - their kernels are small and sit
    // behind list A's issue-bound launches (k_ext_frag<13,25>: 0.03 ms alone, 0.5 ms in the round), which delays the chain that trails list A
    // (= 2: list A's stream instead - its chain is the round's critical one)
    static const int prio = getenv("NECAT_STREAM_PRIO") ? atoi(getenv("NECAT_STREAM_PRIO")) : 0;
    int least = 0, greatest = 0;
    if (prio && hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
    // (the copy stream - deferred column copies of the consensus loop - only for the calls that use it: every stream is a share of the runtime's hardware queues,
    // GPU_MAX_HW_QUEUES, and kernels of streams that share a queue run one after the other; with the second lane's two streams a context has eight)
    for (hipStream_t* st : {&ctx->stream_a, &ctx->stream_b, &ctx->stream_c, &ctx->stream_d, &ctx->stream_copy}) {
        if (*st || (st == &ctx->stream_copy && !with_copy)) continue;
        const bool high = prio && greatest != least && (prio == 2 ? st == &ctx->stream_a : (st == &ctx->stream_b || st == &ctx->stream_c || st == &ctx->stream_d));
        if ((high ? hipStreamCreateWithPriority(st, hipStreamDefault, greatest) : hipStreamCreate(st)) != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "hipStreamCreate failed");
    }
    return NECAT_OK;
}

int extend_impl(necat_ctx* ctx, const necat_volume* ref, const necat_volume* reads, int read_start_id, int ref_start_id,
                const necat_candidate* cands, uint64_t n, const necat_map_options* opt, int tail_match_len,
                necat_m4** out, uint64_t* n_out, AlignOut* ao, const DevCands* dev = nullptr, DevOut* devout = nullptr, RmOut* rm = nullptr)
{
    if (int rc0 = ext_streams(ctx, ao && ao->defer_copy)) return rc0;
    // dev != nullptr (necat_map_pair): the candidates are this library's own, still on the device
    auto t_prev = std::chrono::steady_clock::now();
    auto tick = [&](const char* what) {
        if (!(g_trace & 2)) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[necat] extend %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    if (n >= (1ULL << 31)) return set_err(ctx, NECAT_ERR_ARG, "too many candidates in one call");
    for (uint64_t i = 0; i < (dev ? 0 : n); ++i) {
        const necat_candidate& c = cands[i];
        const int64_t lq = (int64_t)c.qid - read_start_id, ls = (int64_t)c.sid - ref_start_id;
        if (lq < 0 || (uint64_t)lq >= reads->nseq || ls < 0 || (uint64_t)ls >= ref->nseq)
            return set_err(ctx, NECAT_ERR_ARG, "candidate %lu refers to a read outside the volumes", (unsigned long)i);
        if (c.qsize != reads->h_seq_off[lq + 1] - reads->h_seq_off[lq] || c.ssize != ref->h_seq_off[ls + 1] - ref->h_seq_off[ls] ||
            c.qoff > c.qsize || c.soff > c.ssize)
            return set_err(ctx, NECAT_ERR_ARG, "candidate %lu has inconsistent sizes/anchor", (unsigned long)i);
    }
    tick("validate candidates");
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DevVolume dref = dev_view(ref), drd = dev_view(reads);
    ctx->tm.myers_ms = ctx->tm.traceback_ms = 0; ctx->tm.myers_launches = ctx->tm.myers_blocks = ctx->tm.rounds = 0;
    ctx->tm.myers_word_updates = ctx->tm.myers_cells_bases = ctx->tm.myers_band_words = 0;
    ctx->tm.myersA_ms = ctx->tm.tracebackA_ms = 0; ctx->tm.myersA_launches = ctx->tm.myersA_blocks = 0;
    ctx->tm.myersA_big_ms = 0; ctx->tm.myersA_big_blocks = 0;
    ctx->tm.fused_ms = 0; ctx->tm.fused_launches = ctx->tm.fused_blocks = 0;
    ctx->tm.rc_ms = ctx->tm.rc_ck_ms = 0; ctx->tm.rc_launches = ctx->tm.rc_blocks = ctx->tm.rc_words = 0;
    NECAT_HIP(ctx, hipEventRecord(ctx->ev[0], s));
    // batches of <= 786 432 candidates: every batch ends in ~20 latency-bound rounds, so fewer and bigger is better
    // (yeast-size: 654 -> 615 ms against 393 216); their band records need <= 103 GB for list A + a few GB for list B
    // of the 288 GB (NECAT_BATCH overrides)
    // Two lanes (NECAT_EXT_OVERLAP, default on; not in the alignment-keeping mode, whose batches hand columns to the host in between): two batches run their
    // rounds side by side.  A round is a chain of kernels (fragments -> pass -> walk -> finish) each of which drains before the next ramps up - 0.14 + 0.21 ms of a
    // 110 k-block round's 1.0 ms (NOTES_r05 6) - and a batch ends in ~ 15 rounds that are one block's dependent chain each whatever their size; the other lane's
    // kernels fill both.  Yeast size (four batches): 300.7 -> 278 - 283 ms per step.  NECAT_EXT_OVERLAP_MIN > 0 cuts ONE batch of at least that many candidates in
    // two for the same effect (E. coli size, first batch = the 20 % longest chains: 36.8 - 39.6 against 38.8 - 39.3 ms - not a reliable gain, not the default: knobs.h).
    const bool overlap = g_ext_overlap && !ao && !ctx->serial_streams;
    uint64_t n_batches = (n + g_batch_cap - 1) / g_batch_cap;
    if (overlap && n_batches == 1 && g_ext_overlap_min && n >= g_ext_overlap_min) n_batches = 2;
    // batch sizes: equal shares, or - one batch cut in two - NECAT_EXT_OVERLAP_SPLIT per cent (default 20) of the candidates in the first
    std::vector<u32> bsize;
    if (n) {
        const bool cut = overlap && (n + g_batch_cap - 1) / g_batch_cap == 1 && n_batches == 2;
        const u64 share = cut ? std::min<u64>(n, std::max<u64>(64, (n * g_ext_overlap_split / 100 + 63) & ~63ULL)) : (((n + n_batches - 1) / n_batches) + 63) & ~63ULL;
        for (u64 at = 0; at < n;) { const u64 m = std::min<u64>(n - at, cut && at ? n - at : share); bsize.push_back((u32)m); at += m; }
    }
    n_batches = bsize.size();
    const u32 cap = n ? (*std::max_element(bsize.begin(), bsize.end()) + 63) & ~63u : 64u;
    const int nlanes = overlap && n_batches > 1 && g_ext_lanes > 1 ? (int)std::min<uint64_t>(g_ext_lanes, n_batches) : 1;
    const u32 groups = cap / 64 + 1;
    int rc;
    // candidate-wide arrays
    const uint64_t n_groups_max = n;
    const size_t cand_bytes = n * sizeof(necat_candidate) + 2 * n * sizeof(necat_m4) + ((n + 63) & ~63ULL) + (n_groups_max + 1) * 8 + 2048;
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_EXT_CAND], cand_bytes))) return rc;
    char* cb = (char*)ctx->scratch[SC_EXT_CAND].p;
    necat_candidate* d_cands = (necat_candidate*)cb; cb += n * sizeof(necat_candidate);
    necat_m4* d_m4 = (necat_m4*)cb; cb += n * sizeof(necat_m4);
    necat_m4* d_out = (necat_m4*)cb; cb += n * sizeof(necat_m4);
    u64* d_goff = (u64*)cb; cb += (n_groups_max + 1) * 8;
    u32* d_outcnt = (u32*)cb; cb += 1024;         // [0..1] output counter, [2 + 32 l .. 17 + 32 l] list counters (4 buffers x 4) of lane l < kMaxExtLanes
    int* d_err = (int*)cb; cb += 64;
    u8* d_ok = (u8*)cb;
    NECAT_HIP(ctx, hipMemcpyAsync(d_cands, dev ? dev->d : cands, n * sizeof(necat_candidate), dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    static_assert((2 + 32 * (kMaxExtLanes - 1) + 16) * 4 <= 1024, "the lanes' list counters");
    NECAT_HIP(ctx, hipMemsetAsync(d_outcnt, 0, 1024 + 64, s));        // (the counters and the error flag behind them)
    auto cleanup = [&]() {};
    ExtLane lane[kMaxExtLanes];
    for (int l = 0; l < nlanes; ++l) {
        if ((rc = ext_lane(ctx, l, lane[l]))) return rc;
        if ((rc = buf_ensure(ctx, *lane[l].tasks, (size_t)cap * sizeof(ExtTask) + 64)) ||
            (rc = buf_ensure(ctx, *lane[l].lists, (size_t)cap * 10 * sizeof(BlockItem) + 2 * 4096 + 64)) ||
            (rc = buf_ensure(ctx, *lane[l].frag, (size_t)groups * 64 * (kFragWordsA + 2 * kFragWordsB) * 8)) ||
            (rc = buf_ensure(ctx, *lane[l].ops, (size_t)groups * 64 * (kOpsA + 2 * kOpsB))) ||
            (rc = buf_ensure(ctx, *lane[l].res, (size_t)groups * 64 * 3 * sizeof(BlockResult)))) { cleanup(); return rc; }
    }
    // Several batches: every batch runs as many rounds as its longest chain of blocks and ends in latency-bound rounds,
    // so the candidates are dealt to the batches by expected chain length (what is left of the two reads beyond the
    // anchor, in blocks), longest first: the first batch has the ~30-round chains, the last ones a handful of rounds.
    u32* d_perm = nullptr;
    if (n_batches > 1 && !ao && g_ext_overlap_order) {
        // on the device (k_len_order): the candidates may never have been on the host (necat_map_pair), and a host counting sort of
        // millions of 88-byte records costs more than a batch's first rounds
        if ((rc = buf_ensure(ctx, ctx->scratch[SC_EXT_PERM], n * 4 + 2 * kLenBins * 4 + 64))) { cleanup(); return rc; }
        d_perm = (u32*)ctx->scratch[SC_EXT_PERM].p;
        u32* d_cur = d_perm + n;
        NECAT_HIP(ctx, hipMemsetAsync(d_cur, 0, kLenBins * 4, s));
        hipLaunchKernelGGL(k_len_order<0>, dim3(grid_for(n, 256, 1u << 23)), dim3(256), 0, s, (const necat_candidate*)d_cands, (u32)n, d_cur, (u32*)nullptr);
        u32 cnt[kLenBins], start[kLenBins];
        NECAT_HIP(ctx, hipMemcpyAsync(cnt, d_cur, sizeof cnt, hipMemcpyDeviceToHost, s));
        NECAT_HIP(ctx, hipStreamSynchronize(s));
        u32 run = 0;
        for (int b = 0; b < kLenBins; ++b) { start[b] = run; run += cnt[b]; }
        NECAT_HIP(ctx, hipMemcpyAsync(d_cur, start, sizeof start, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_len_order<1>, dim3(grid_for(n, 256, 1u << 23)), dim3(256), 0, s, (const necat_candidate*)d_cands, (u32)n, d_cur, d_perm);
        NECAT_CHECK_LAUNCH(ctx, "k_len_order");
        NECAT_HIP(ctx, hipStreamSynchronize(s));       // `start` is a local
    }
    NECAT_HIP(ctx, hipStreamSynchronize(s));        // candidates + zeroed counters are in place before the batch streams start
    tick("buffers + upload");
    Batch kb[kMaxExtLanes];
    for (int l = 0; l < nlanes; ++l) {
        Batch& k = kb[l]; const ExtLane& E = lane[l];
        k.tasks = (ExtTask*)E.tasks->p;
        BlockItem* q = (BlockItem*)E.lists->p;
        for (int j = 0; j < 4; ++j) { k.itemsA[j] = q + (size_t)(2 * j) * cap; k.itemsB[j] = q + (size_t)(2 * j + 1) * cap; }
        k.fragA = (u64*)E.frag->p;
        k.opsA = (u8*)E.ops->p;
        k.resA = (BlockResult*)E.res->p;
        for (int j = 0; j < 2; ++j) {
            k.sortedB[j] = q + (size_t)(8 + j) * cap; k.bins[j] = (u32*)(q + 10 * (size_t)cap) + 1024 * j;
            k.fragB[j] = k.fragA + (size_t)groups * 64 * (kFragWordsA + j * kFragWordsB);
            k.opsB[j] = k.opsA + (size_t)groups * 64 * (kOpsA + j * kOpsB);
            k.resB[j] = k.resA + (size_t)groups * 64 * (1 + j);
        }
        k.count = d_outcnt + 2 + 32 * l; k.cap = cap;
        k.sa = E.sa; k.sb[0] = E.sb[0]; k.sb[1] = E.sb[1];
        for (int j = 0; j < 4; ++j) { k.a0[j] = E.ev[4 + 3 * j]; k.a1[j] = E.ev[5 + 3 * j]; k.a2[j] = E.ev[6 + 3 * j]; }     // ev[4..15]
        for (int j = 0; j < 2; ++j) { k.b0[j] = E.ev[18 + 3 * j]; k.b1[j] = E.ev[19 + 3 * j]; k.b2[j] = E.ev[20 + 3 * j]; } // ev[18..23]
        NECAT_HIP(ctx, hipMemsetAsync(k.bins[0], 0, 2 * 4096, k.sa));     // size-sort counters of list B: reset by the kernels after every use
        k.base = 0; k.n = 0;
    }
    Batch& k = kb[0];
    ExtShared X;
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_STATS], kStatBytes))) return rc;          // the work counters, kStatSlots copies (stat_add, ext_kernels.h)
    NECAT_HIP(ctx, hipMemsetAsync(ctx->scratch[SC_STATS].p, 0, kStatBytes, s));
    X.d_cands = d_cands; X.d_m4 = d_m4; X.d_ok = d_ok; X.d_err = d_err; X.stats = (unsigned long long*)ctx->scratch[SC_STATS].p;
    X.error = opt->error; X.tail_match_len = tail_match_len; X.min_align = opt->align_size_cutoff;
    X.read_start_id = read_start_id; X.ref_start_id = ref_start_id; X.reads_off = reads->seq_off; X.ref_off = ref->seq_off;
    std::vector<u64> goff;
    if (nlanes >= 2) {
        // ---- two lanes: batch i + 1 starts on the free lane once batch i is in its tail (BatchRun::tail); ONE host thread turns both round loops,
        // whichever has its next list sizes published (BatchRun::ready) - the host still never waits for the device inside a loop
        struct LaneRun { std::unique_ptr<BatchRun> run; int state = 0; int rc = NECAT_OK; u32 polls = 0; };      // state: 0 free, 1 in its rounds, 2 draining, 3 its result kernel in flight
        LaneRun lr[kMaxExtLanes];
        uint64_t next_base = 0, done = 0; size_t started = 0;
        int last = -1;                              // the lane of the batch started last
        auto start = [&](int l) -> int {
            Batch& b = kb[l];
            b.base = next_base; b.n = bsize[started++]; next_base += b.n;
            NECAT_HIP(ctx, hipMemsetAsync(b.count, 0, 64, b.sa));
            ExtLists L0; L0.count = b.count; L0.itemsA = b.itemsA[0]; L0.itemsB = b.itemsB[0]; L0.capA = cap;
            hipLaunchKernelGGL(k_ext_init, dim3(grid_for(b.n, 256)), dim3(256), 0, b.sa, (const necat_candidate*)d_cands, b.n, (u32)b.base,
                               read_start_id, ref_start_id, X.reads_off, X.ref_off, b.tasks, L0, (const u64*)nullptr, (const u32*)d_perm, rm ? 1 : 0);
            NECAT_CHECK_LAUNCH(ctx, "k_ext_init");
            lr[l].run.reset(new BatchRun(ctx, dref, drd, b, X, lane[l])); lr[l].state = 1; lr[l].rc = NECAT_OK; last = l;
            if (g_trace & 1) fprintf(stderr, "[necat] batch@%lu (%u candidates) starts on lane %d\n", (unsigned long)b.base, b.n, l);
            return NECAT_OK;
        };
        int err = NECAT_OK;
        u64 idle = 0; double t_idle = wall_ms();
        while (done < n_batches && !err) {
            bool progressed = false;
            if (next_base < n && (last < 0 || lr[last].state != 1 || lr[last].run->tail || g_ext_overlap_pct >= 100)) {
                for (int l = 0; l < nlanes; ++l) if (lr[l].state == 0) {
                    if ((err = start(l))) break;
                    progressed = true;
                    if (goff.empty() && dev) goff = dev->group_off;
                    if (goff.empty()) {
                        // while the first kernels run: groups of equal qid for the containment filter (candidates arrive grouped per read: pm_worker.c:100-140)
                        goff.push_back(0);
                        for (uint64_t i = 1; i < n; ++i) if (cands[i].qid != cands[i - 1].qid) goff.push_back(i);
                        goff.push_back(n);
                    }
                    break;
                }
                if (err) break;
            }
            for (int l = 0; l < nlanes && !err; ++l) {
                LaneRun& R = lr[l];
                if (R.state == 1 && R.run->ready()) { R.rc = R.run->step(); progressed = true; if (R.run->over) R.state = 2; }
                // (a draining lane is looked at every 64th turn of the loop: three hipStreamQuery calls per turn would be the OTHER lane's launch latency)
                if (R.state == 2 && (R.rc || ((++R.polls & 63u) == 0 && R.run->drained()))) {
                    if (!(err = R.run->finish(R.rc))) {
                        // the batch's records: launched and left to an event - the host thread goes on turning the other lane's rounds instead of waiting here
                        hipLaunchKernelGGL(k_ext_result, dim3(grid_for(kb[l].n, 256)), dim3(256), 0, kb[l].sa, (const ExtTask*)kb[l].tasks, kb[l].n, (const necat_candidate*)d_cands,
                                           opt->align_size_cutoff, d_m4, d_ok, rm ? 1 : 0);
                        if (hipGetLastError() != hipSuccess || hipEventRecord(lane[l].ev[31], kb[l].sa) != hipSuccess) err = set_err(ctx, NECAT_ERR_DEVICE, "k_ext_result failed");
                    }
                    R.run.reset(); R.state = err ? 0 : 3; R.polls = 0; progressed = true;
                    if (err) ++done;
                }
                if (R.state == 3 && (++R.polls & 15u) == 0) {
                    const hipError_t q = hipEventQuery(lane[l].ev[31]);
                    if (q == hipErrorNotReady) (void)hipGetLastError();
                    else {
                        if (q != hipSuccess) err = set_err(ctx, NECAT_ERR_DEVICE, "k_ext_result failed: %s", hipGetErrorString(q));
                        R.state = 0; ++done; progressed = true;
                    }
                }
            }
            if (progressed) { idle = 0; t_idle = wall_ms(); continue; }
            if ((++idle & 0xfffff) == 0) {
                // a failed kernel never publishes: look at the streams instead of spinning forever
                for (int l = 0; l < nlanes && !err; ++l) if (lr[l].state == 1) {
                    const hipError_t q = hipStreamQuery(kb[l].sa);
                    if (q != hipSuccess && q != hipErrorNotReady) err = set_err(ctx, NECAT_ERR_DEVICE, "extension rounds (lane %d) failed: %s", l, hipGetErrorString(q));
                }
                if (!err && wall_ms() - t_idle > 120e3) err = set_err(ctx, NECAT_ERR_DEVICE, "extension rounds: no progress for 120 s");
            }
        }
        if (err) {
            for (int l = 0; l < nlanes; ++l) { LaneRun& R = lr[l]; if (R.state == 3) (void)hipStreamSynchronize(kb[l].sa); else if (R.state) { (void)R.run->finish(err); R.run.reset(); } }       // nothing of a lane is in flight when its buffers are handed on
            cleanup(); return err;
        }
    } else
    for (uint64_t next_base = 0, bi = 0; next_base < n; ++bi) {
        k.base = next_base; k.n = bsize[bi]; next_base += k.n;
        NECAT_HIP(ctx, hipMemsetAsync(k.count, 0, 64, k.sa));
        ExtLists L0; L0.count = k.count; L0.itemsA = k.itemsA[0]; L0.itemsB = k.itemsB[0]; L0.capA = cap;
        const u64* d_ops_base = nullptr;
        if (ao) {
            // column region of a task: left stream (<= qoff + soff columns) then right stream
            // (<= what is left of both reads from the anchor the left extension moved back)
            std::vector<u64> base(k.n + 1, 0);
            for (u32 i = 0; i < k.n; ++i) {
                const necat_candidate& c = cands[k.base + i];
                base[i + 1] = base[i] + ((c.qsize + c.ssize + c.qoff + c.soff + 64) / 32 + 2) * 8;      // bytes: 2 bits per column
            }
            if ((rc = buf_ensure(ctx, ctx->scratch[SC_EXT_COLS], base[k.n] + (size_t)(k.n + 1) * 8 + 64))) { cleanup(); return rc; }
            X.task_ops = (u8*)ctx->scratch[SC_EXT_COLS].p;
            u64* d_base = (u64*)(X.task_ops + ((base[k.n] + 63) & ~63ULL));
            NECAT_HIP(ctx, hipMemcpyAsync(d_base, base.data(), (size_t)k.n * 8, hipMemcpyHostToDevice, k.sa));
            NECAT_HIP(ctx, hipStreamSynchronize(k.sa));
            d_ops_base = d_base;
        }
        hipLaunchKernelGGL(k_ext_init, dim3(grid_for(k.n, 256)), dim3(256), 0, k.sa, (const necat_candidate*)d_cands, k.n, (u32)k.base,
                           read_start_id, ref_start_id, X.reads_off, X.ref_off, k.tasks, L0, d_ops_base, (const u32*)d_perm, rm ? 1 : 0);
        NECAT_CHECK_LAUNCH(ctx, "k_ext_init");
        if (goff.empty() && dev) goff = dev->group_off;
        if (goff.empty() && !ao) {
            // while the first kernels run: groups of equal qid for the containment filter
            // (candidates arrive grouped per read: pm_worker.c:100-140)
            goff.push_back(0);
            for (uint64_t i = 1; i < n; ++i) if (cands[i].qid != cands[i - 1].qid) goff.push_back(i);
            goff.push_back(n);
        }
        { BatchRun run(ctx, dref, drd, k, X, lane[0]); if ((rc = run.run())) { cleanup(); return rc; } }
        if (!ao) {
            hipLaunchKernelGGL(k_ext_result, dim3(grid_for(k.n, 256)), dim3(256), 0, k.sa, (const ExtTask*)k.tasks, k.n, (const necat_candidate*)d_cands,
                               opt->align_size_cutoff, d_m4, d_ok, rm ? 1 : 0);
            NECAT_CHECK_LAUNCH(ctx, "k_ext_result");
            NECAT_HIP(ctx, hipStreamSynchronize(k.sa));
        } else {
            // per-candidate results + the batch's alignment columns, packed in candidate order
            necat_alignment* d_aln = (necat_alignment*)d_m4;          // the M4 arrays are not used in this mode
            u32* d_len = (u32*)d_out;
            hipLaunchKernelGGL(k_ext_alignment, dim3(grid_for(k.n, 256)), dim3(256), 0, k.sa, (const ExtTask*)k.tasks, k.n, 0u,
                               opt->align_size_cutoff, d_aln, d_len);
            NECAT_CHECK_LAUNCH(ctx, "k_ext_alignment");
            std::vector<u32> len(k.n);
            NECAT_HIP(ctx, hipMemcpyAsync(len.data(), d_len, (size_t)k.n * 4, hipMemcpyDeviceToHost, k.sa));
            NECAT_HIP(ctx, hipMemcpyAsync(ao->aln + k.base, d_aln, (size_t)k.n * sizeof(necat_alignment), hipMemcpyDeviceToHost, k.sa));
            NECAT_HIP(ctx, hipStreamSynchronize(k.sa));
            // every alignment starts on a 64-bit word: 32 columns per word
            std::vector<u64> off(k.n + 1, 0);
            for (u32 i = 0; i < k.n; ++i) off[i + 1] = off[i] + (len[i] + 31) / 32;
            const u64 tot = off[k.n] * 8, at = ao->total;
            for (u32 i = 0; i < k.n; ++i) ao->off[k.base + i] = at + off[i] * 8;
            ao->off[k.base + k.n] = at + tot;
            if (tot) {
                const size_t need_out = tot + (size_t)(k.n + 1) * 8 + 64;
                if (ctx->copy_pending && need_out > ctx->scratch[SC_EXT_COLS_OUT].cap) {      // the buffer is about to be replaced
                    NECAT_HIP(ctx, hipStreamSynchronize(ctx->stream_copy)); ctx->copy_pending = false;
                }
                if ((rc = buf_ensure(ctx, ctx->scratch[SC_EXT_COLS_OUT], need_out))) { cleanup(); return rc; }
                if (ctx->copy_pending) { NECAT_HIP(ctx, hipStreamWaitEvent(k.sa, ctx->ev[17], 0)); ctx->copy_pending = false; }
                u8* d_cols = (u8*)ctx->scratch[SC_EXT_COLS_OUT].p;
                u64* d_off = (u64*)(d_cols + ((tot + 63) & ~63ULL));
                NECAT_HIP(ctx, hipMemcpyAsync(d_off, off.data(), (size_t)k.n * 8, hipMemcpyHostToDevice, k.sa));
                hipLaunchKernelGGL(k_ext_strings, dim3(grid_for((u64)k.n * 64, 256)), dim3(256), 0, k.sa, (const ExtTask*)k.tasks, k.n,
                                   (const u8*)X.task_ops, (const u64*)d_off, (u64*)d_cols);
                NECAT_CHECK_LAUNCH(ctx, "k_ext_strings");
                u8* part = (u8*)result_alloc(tot);
                if (!part) { cleanup(); return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"); }
                ao->parts.emplace_back(part, tot); ao->total += tot;
                if (ao->defer_copy) {
                    NECAT_HIP(ctx, hipEventRecord(ctx->ev[16], k.sa));
                    NECAT_HIP(ctx, hipStreamWaitEvent(ctx->stream_copy, ctx->ev[16], 0));
                    NECAT_HIP(ctx, hipMemcpyAsync(part, d_cols, tot, hipMemcpyDeviceToHost, ctx->stream_copy));
                    NECAT_HIP(ctx, hipEventRecord(ctx->ev[17], ctx->stream_copy));
                    ctx->copy_pending = true;
                    NECAT_HIP(ctx, hipStreamSynchronize(k.sa));      // the batch's kernels are done (its buffers are reused next)
                } else {
                    NECAT_HIP(ctx, hipMemcpyAsync(part, d_cols, tot, hipMemcpyDeviceToHost, k.sa));
                    NECAT_HIP(ctx, hipStreamSynchronize(k.sa));
                }
            }
        }
    }
    tick("rounds");
    {
        unsigned long long hs[5] = {0, 0, 0, 0, 0};
        std::vector<unsigned long long> copies((size_t)kStatSlots * kStatStride);
        NECAT_HIP(ctx, hipMemcpyAsync(copies.data(), X.stats, kStatBytes, hipMemcpyDeviceToHost, s));
        NECAT_HIP(ctx, hipStreamSynchronize(s));
        for (int c = 0; c < kStatSlots; ++c) for (int q = 0; q < 5; ++q) hs[q] += copies[(size_t)c * kStatStride + q];
        ctx->tm.myers_word_updates = hs[0]; ctx->tm.myers_cells_bases = hs[1]; ctx->tm.myers_band_words = hs[2];
        ctx->tm.rc_blocks = hs[3]; ctx->tm.rc_words = hs[4];
    }
    if (ao) {
        int herr = 0;
        NECAT_HIP(ctx, hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, s));
        NECAT_HIP(ctx, hipEventRecord(ctx->ev[1], s));
        NECAT_HIP(ctx, hipStreamSynchronize(s));
        ctx->tm.extend_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
        if (herr) return set_err(ctx, NECAT_ERR_INTERNAL, "extension kernels reported error code %d", herr);
        return NECAT_OK;
    }
    if (rm) {
        int herr = 0;
        rm->cands.resize(n); rm->m4.resize(n); rm->ok.resize(n); rm->group_off = goff;
        NECAT_HIP(ctx, hipMemcpyAsync(rm->cands.data(), d_cands, n * sizeof(necat_candidate), hipMemcpyDeviceToHost, s));
        NECAT_HIP(ctx, hipMemcpyAsync(rm->m4.data(), d_m4, n * sizeof(necat_m4), hipMemcpyDeviceToHost, s));
        NECAT_HIP(ctx, hipMemcpyAsync(rm->ok.data(), d_ok, n, hipMemcpyDeviceToHost, s));
        NECAT_HIP(ctx, hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, s));
        NECAT_HIP(ctx, hipEventRecord(ctx->ev[1], s));
        NECAT_HIP(ctx, hipStreamSynchronize(s));
        ctx->tm.extend_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
        cleanup();
        if (herr) return set_err(ctx, NECAT_ERR_INTERNAL, "extension kernels reported error code %d", herr);
        return NECAT_OK;
    }
    const u32 ng = (u32)goff.size() - 1;
    NECAT_HIP(ctx, hipMemcpyAsync(d_goff, goff.data(), goff.size() * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_m4_filter, dim3(grid_for((u64)ng * 64, 256)), dim3(256), 0, s, (const necat_candidate*)d_cands, (const u64*)d_goff, ng,
                       (const necat_m4*)d_m4, d_ok, d_out, d_outcnt);
    NECAT_CHECK_LAUNCH(ctx, "k_m4_filter");
    u32 nout = 0; int herr = 0;
    NECAT_HIP(ctx, hipMemcpyAsync(&nout, d_outcnt, 4, hipMemcpyDeviceToHost, s));
    NECAT_HIP(ctx, hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, s));
    NECAT_HIP(ctx, hipStreamSynchronize(s));
    if (herr) { cleanup(); return set_err(ctx, NECAT_ERR_INTERNAL, "extension kernels reported error code %d", herr); }
    if (devout) {
        devout->d = d_out; devout->n = nout;
        NECAT_HIP(ctx, hipEventRecord(ctx->ev[1], s));
        NECAT_HIP(ctx, hipStreamSynchronize(s));
        ctx->tm.extend_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
        return NECAT_OK;
    }
    tick("filter");
    necat_m4* res = (necat_m4*)result_alloc(std::max<size_t>(1, nout) * sizeof(necat_m4));
    if (!res) { cleanup(); return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"); }
    tick("result block");
    if (nout) NECAT_HIP(ctx, hipMemcpyAsync(res, d_out, (size_t)nout * sizeof(necat_m4), hipMemcpyDeviceToHost, s));
    NECAT_HIP(ctx, hipEventRecord(ctx->ev[1], s));
    NECAT_HIP(ctx, hipStreamSynchronize(s));
    ctx->tm.extend_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
    tick("copy to host");
    cleanup();
    *out = res; *n_out = nout;
    return NECAT_OK;
}
}  // namespace

int necat_extend(necat_ctx* ctx, const necat_volume* ref, const necat_volume* reads, int read_start_id, int ref_start_id,
                 const necat_candidate* cands, uint64_t n, const necat_map_options* opt, int tail_match_len,
                 necat_m4** out, uint64_t* n_out)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ref || !reads || !opt || !out || !n_out || (n && !cands)) return NECAT_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (n == 0) return NECAT_OK;
    return extend_impl(ctx, ref, reads, read_start_id, ref_start_id, cands, n, opt, tail_match_len, out, n_out, nullptr);
}

int necat_map_pair(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                   int read_start_id, int ref_start_id, int pairwise, const necat_map_options* opt, int tail_match_len,
                   necat_m4** out, uint64_t* n_out, uint64_t* n_candidates)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ix || !ref || !reads || !opt || !out || !n_out) return NECAT_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (n_candidates) *n_candidates = 0;
    necat_map_options o = *opt;
    o.job = 1;                                   // the candidates of a mapping job: always sorted, cut to num_candidates
    DevCands dev;
    int rc = find_impl(ctx, ix, ref, reads, read_start_id, ref_start_id, pairwise, &o, nullptr, nullptr, &dev);
    if (rc) return rc;
    if (n_candidates) *n_candidates = dev.n;
    if (dev.n == 0) return NECAT_OK;
    return extend_impl(ctx, ref, reads, read_start_id, ref_start_id, nullptr, dev.n, &o, tail_match_len, out, n_out, nullptr, &dev);
}
