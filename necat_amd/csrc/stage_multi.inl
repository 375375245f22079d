// stage_multi.inl - one volume on several GPUs, a share of a pair, the pair schedule (comm.h, pair_sched.h).
// One of the stage files of libnecat_hip.so's single translation unit: necat_hip.hip includes them in order, inside its extern "C" block, after the
// context / knob / result-pool code they all use (the kernels are header templates and the stages share host helpers: one device code object, one 30 s build).

// ------------------------------------------------------------------------------------------ one volume on several GPUs

int necat_comm_create(necat_ctx* ctx, int rank, int nranks, necat_host_allgather_fn fn, void* user, const char* transport, necat_comm** out)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !fn)) return NECAT_ERR_ARG;
    *out = nullptr;
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    necat_comm* c = new necat_comm();
    c->rank = rank; c->nranks = nranks; c->gather = fn; c->user = user;
    int want = -1;                                   // -1 = auto
    const char* env = getenv("NECAT_COMM");
    const char* t = (transport && *transport && strcmp(transport, "auto")) ? transport : (env && *env ? env : "auto");
    if (!strcmp(t, "rccl")) want = 0; else if (!strcmp(t, "ipc")) want = 1;
    else if (strcmp(t, "auto")) { delete c; return set_err(ctx, NECAT_ERR_ARG, "unknown transport '%s' (auto, rccl, ipc)", t); }
    int rc = NECAT_OK;
    if (nranks > 1) {
        // who shares a device with whom: (host, PCI bus id) of every rank
        struct Where { char host[64]; char bus[32]; } me, *all;
        std::vector<Where> ws(nranks);
        all = ws.data();
        memset(&me, 0, sizeof me);
        (void)gethostname(me.host, sizeof me.host - 1);
        if (hipDeviceGetPCIBusId(me.bus, sizeof me.bus, ctx->device) != hipSuccess) snprintf(me.bus, sizeof me.bus, "dev%d", ctx->device);
        if ((rc = comm::host_allgather(ctx, c, &me, all, sizeof(Where)))) { delete c; return rc; }
        bool shared = false;
        for (int a = 0; a < nranks; ++a) for (int b = a + 1; b < nranks; ++b)
            if (!strcmp(all[a].host, all[b].host) && !strcmp(all[a].bus, all[b].bus)) shared = true;
        if (want < 0) want = shared ? 1 : 0;
        if (want == 0 && shared) { delete c; return set_err(ctx, NECAT_ERR_COMM, "RCCL cannot run two ranks on one device (use transport \"ipc\")"); }
    }
    const bool was_auto = want < 0 || !strcmp(t, "auto");
    c->transport = want < 0 ? 0 : want;
    if (c->transport == 0 && nranks > 1) {
        // bring RCCL up; every rank reports, and with transport "auto" ANY failure sends all ranks to the IPC transport
        // (device-to-device copies through HIP IPC handles: the same pull pattern, xGMI underneath) instead of failing the job
        int ok = comm::load_rccl(ctx, c) == NECAT_OK;
        ncclUniqueId id;
        std::vector<ncclUniqueId> all_ids(nranks);
        memset(&id, 0, sizeof id);
        if (ok && rank == 0) { const ncclResult_t r = c->p_GetUniqueId(&id); if (r != ncclSuccess) { set_err(ctx, NECAT_ERR_COMM, "ncclGetUniqueId: %s", c->p_GetErrorString(r)); ok = 0; } }
        if ((rc = comm::host_allgather(ctx, c, &id, all_ids.data(), sizeof id))) { delete c; return rc; }
        std::vector<int> oks(nranks, 0);
        if ((rc = comm::host_allgather(ctx, c, &ok, oks.data(), sizeof(int)))) { delete c; return rc; }
        bool all_ok = true; for (int v : oks) all_ok = all_ok && v;
        if (all_ok) {
            const ncclResult_t r = c->p_CommInitRank(&c->nccl, nranks, all_ids[0], rank);
            if (r != ncclSuccess) { set_err(ctx, NECAT_ERR_COMM, "ncclCommInitRank: %s", c->p_GetErrorString(r)); ok = 0; c->nccl = nullptr; }
            if ((rc = comm::host_allgather(ctx, c, &ok, oks.data(), sizeof(int)))) { delete c; return rc; }
            all_ok = true; for (int v : oks) all_ok = all_ok && v;
        }
        if (all_ok) {
            // first contact: a ring of 4 KB messages over the new communicator, checked byte for byte (comm::first_contact) - the ranks sit on
            // distinct devices here, and an exchange between two devices is exactly what no single-GPU box ever ran
            constexpr size_t kContact = 4096;
            void* d = nullptr;
            int fc = hipMalloc(&d, 2 * kContact) == hipSuccess ? NECAT_OK : set_err(ctx, NECAT_ERR_MEMORY, "hipMalloc (first contact)");
            if (fc) { (void)hipGetLastError(); fc = comm::agree(ctx, c, fc); } else fc = comm::first_contact(ctx, c, d, kContact, ctx->stream);
            if (d) (void)hipFree(d);
            if (fc) { all_ok = false; ok = 0; }
            else if (g_trace && rank == 0) fprintf(stderr, "[necat] RCCL first contact among %d ranks: ok\n", nranks);
        }
        if (!all_ok) {
            if (c->nccl && c->p_CommDestroy) { (void)c->p_CommDestroy(c->nccl); c->nccl = nullptr; }
            if (!was_auto) { const int e = ok ? set_err(ctx, NECAT_ERR_COMM, "RCCL could not be initialised on another rank") : NECAT_ERR_COMM; delete c; return e; }      // (ctx->err: this rank's RCCL / HIP error text)
            if (rank == 0) fprintf(stderr, "[necat] RCCL transport unavailable (%s): using HIP IPC copies\n", ok ? "another rank failed" : ctx->err);
            c->transport = 1;
        }
    }
    *out = c;
    return NECAT_OK;
}

// Test hook: the RCCL transport's whole call path in ONE process - librccl opened at run time, a communicator of one rank, a
// send/recv group to itself on the context's stream - so that it runs on hardware even where a second GPU is not available.
int necat_comm_selftest_rccl(necat_ctx* ctx, uint64_t bytes)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !bytes) return NECAT_ERR_ARG;
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    necat_comm c;
    c.rank = 0; c.nranks = 1;
    int rc = comm::load_rccl(ctx, &c);
    if (rc) return rc;
    ncclUniqueId id;
    NECAT_NCCL(ctx, &c, c.p_GetUniqueId(&id));
    NECAT_NCCL(ctx, &c, c.p_CommInitRank(&c.nccl, 1, id, 0));
    unsigned char *a = nullptr, *b = nullptr;
    std::vector<unsigned char> h(bytes), g(bytes);
    for (uint64_t i = 0; i < bytes; ++i) h[i] = (unsigned char)(i * 131u + 7u);
    auto body = [&]() -> int {
        NECAT_HIP(ctx, hipMalloc((void**)&a, bytes)); NECAT_HIP(ctx, hipMalloc((void**)&b, bytes));
        NECAT_HIP(ctx, hipMemcpyAsync(a, h.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
        NECAT_HIP(ctx, hipMemsetAsync(b, 0, bytes, ctx->stream));
        NECAT_NCCL(ctx, &c, c.p_GroupStart());
        NECAT_NCCL(ctx, &c, c.p_Send(a, bytes, ncclChar, 0, c.nccl, ctx->stream));
        NECAT_NCCL(ctx, &c, c.p_Recv(b, bytes, ncclChar, 0, c.nccl, ctx->stream));
        NECAT_NCCL(ctx, &c, c.p_GroupEnd());
        NECAT_HIP(ctx, hipMemcpyAsync(g.data(), b, bytes, hipMemcpyDeviceToHost, ctx->stream));
        NECAT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return memcmp(h.data(), g.data(), bytes) ? set_err(ctx, NECAT_ERR_COMM, "RCCL self send/recv returned different bytes") : NECAT_OK;
    };
    rc = body();
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    (void)c.p_CommDestroy(c.nccl);
    return rc;
}

// Two ranks on two devices in ONE process: the all-pairs exchange of comm.h's RCCL branch in its smallest form - each rank sends its
// buffer to the other and receives the other's, one ncclSend / ncclRecv group per rank, both inside one ncclGroup (as several communicators
// of one process must be driven).  Returns 1 when the box has fewer than two devices (callers skip), 0 when both ranks received the
// right bytes over the link.
int necat_comm_selftest_rccl2(necat_ctx* ctx, uint64_t bytes)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !bytes) return NECAT_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 2) { (void)hipGetLastError(); set_err(ctx, NECAT_OK, "fewer than two devices: the two-rank RCCL exchange cannot run here"); return 1; }
    necat_comm c;
    c.rank = 0; c.nranks = 2;
    int rc = comm::load_rccl(ctx, &c);
    if (rc) return rc;
    const int dev[2] = {ctx->device, ctx->device == 0 ? 1 : 0};
    ncclUniqueId id;
    NECAT_NCCL(ctx, &c, c.p_GetUniqueId(&id));
    ncclComm_t cm[2] = {nullptr, nullptr};
    hipStream_t st[2] = {nullptr, nullptr};
    unsigned char* buf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    std::vector<unsigned char> h[2], g[2];
    for (int r = 0; r < 2; ++r) { h[r].resize(bytes); g[r].resize(bytes); for (uint64_t i = 0; i < bytes; ++i) h[r][i] = (unsigned char)(i * 131u + 7u + 101u * (unsigned)r); }
    auto body = [&]() -> int {
        NECAT_NCCL(ctx, &c, c.p_GroupStart());
        for (int r = 0; r < 2; ++r) { NECAT_HIP(ctx, hipSetDevice(dev[r])); NECAT_NCCL(ctx, &c, c.p_CommInitRank(&cm[r], 2, id, r)); }
        NECAT_NCCL(ctx, &c, c.p_GroupEnd());
        for (int r = 0; r < 2; ++r) {
            NECAT_HIP(ctx, hipSetDevice(dev[r]));
            NECAT_HIP(ctx, hipStreamCreate(&st[r]));
            NECAT_HIP(ctx, hipMalloc((void**)&buf[r][0], bytes)); NECAT_HIP(ctx, hipMalloc((void**)&buf[r][1], bytes));
            NECAT_HIP(ctx, hipMemcpyAsync(buf[r][0], h[r].data(), bytes, hipMemcpyHostToDevice, st[r]));
            NECAT_HIP(ctx, hipMemsetAsync(buf[r][1], 0, bytes, st[r]));
        }
        NECAT_NCCL(ctx, &c, c.p_GroupStart());
        for (int r = 0; r < 2; ++r) {
            NECAT_HIP(ctx, hipSetDevice(dev[r]));
            NECAT_NCCL(ctx, &c, c.p_Send(buf[r][0], bytes, ncclChar, 1 - r, cm[r], st[r]));
            NECAT_NCCL(ctx, &c, c.p_Recv(buf[r][1], bytes, ncclChar, 1 - r, cm[r], st[r]));
        }
        NECAT_NCCL(ctx, &c, c.p_GroupEnd());
        for (int r = 0; r < 2; ++r) {
            NECAT_HIP(ctx, hipSetDevice(dev[r]));
            NECAT_HIP(ctx, hipMemcpyAsync(g[r].data(), buf[r][1], bytes, hipMemcpyDeviceToHost, st[r]));
            NECAT_HIP(ctx, hipStreamSynchronize(st[r]));
        }
        for (int r = 0; r < 2; ++r) if (memcmp(g[r].data(), h[1 - r].data(), bytes)) return set_err(ctx, NECAT_ERR_COMM, "RCCL exchange between devices %d and %d: rank %d received different bytes", dev[0], dev[1], r);
        return NECAT_OK;
    };
    rc = body();
    for (int r = 0; r < 2; ++r) {
        (void)hipSetDevice(dev[r]);
        for (int q = 0; q < 2; ++q) if (buf[r][q]) (void)hipFree(buf[r][q]);
        if (st[r]) (void)hipStreamDestroy(st[r]);
        if (cm[r]) (void)c.p_CommDestroy(cm[r]);
    }
    (void)hipSetDevice(ctx->device);
    return rc;
}

void necat_comm_destroy(necat_comm* c)
{
    if (!c) return;
    if (c->nccl && c->p_CommDestroy) (void)c->p_CommDestroy(c->nccl);
    delete c;                                       // librccl stays loaded: other users in the process may share it
}

int necat_comm_transport(const necat_comm* c, char* buf, size_t n)
{
    if (!c || !buf || !n) return NECAT_ERR_ARG;
    snprintf(buf, n, "%s", c->transport == 0 ? "rccl" : "ipc");
    return NECAT_OK;
}

int necat_get_shard_timings(const necat_ctx* ctx, necat_shard_timings* t)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !t) return NECAT_ERR_ARG;
    *t = ctx->shard_tm;
    return NECAT_OK;
}

int necat_get_shard_timings_sized(const necat_ctx* ctx, void* t, size_t bytes)
{
    if (!ctx || !t) return NECAT_ERR_ARG;
    memcpy(t, &ctx->shard_tm, bytes < sizeof(necat_shard_timings) ? bytes : sizeof(necat_shard_timings));
    return NECAT_OK;
}

namespace {
// gather-v of fixed-size records on `root`: every rank's `n_local` records at d_local (device memory; may be null when 0).
// Root: host_out = all records (its own first), *n_out their number; other ranks: their own records.
// `status`: what this rank's part of the job returned.  A rank that failed still joins the count exchange, with a sentinel count, so
// that every rank leaves together with an error instead of waiting for records that never come.
int gather_records(necat_ctx* ctx, necat_comm* comm, int root, const void* d_local, uint64_t n_local, size_t rec, void** host_out, uint64_t* n_out, int status)
{
    hipStream_t s = ctx->stream;
    const int G = comm->nranks;
    std::vector<unsigned long long> cnt(G, 0);
    const unsigned long long mine = status ? ~0ULL : n_local;
    int rc = comm::host_allgather(ctx, comm, &mine, cnt.data(), 8);
    if (status) return status;
    if (rc) return rc;
    for (int g = 0; g < G; ++g) if (cnt[g] == ~0ULL) return set_err(ctx, NECAT_ERR_COMM, "rank %d failed in its share of the job: no records are gathered", g);
    // the root's own records come first in its output
    std::vector<size_t> bytes(G);
    uint64_t total = 0;
    for (int g = 0; g < G; ++g) { bytes[g] = (size_t)cnt[g] * rec; total += cnt[g]; }
    const bool is_root = comm->rank == root;
    void* d_all = nullptr;
    if (is_root) {
        rc = buf_ensure(ctx, ctx->scratch[SC_GATHER], std::max<size_t>(256, (size_t)total * rec));
        d_all = ctx->scratch[SC_GATHER].p;
    }
    if ((rc = comm::agree(ctx, comm, rc))) return rc;          // the root has its receive buffer, or nobody sends
    if ((rc = comm::agree(ctx, comm, comm::gatherv(ctx, comm, d_local, bytes, root, d_all, s)))) return rc;
    ctx->shard_tm.gather_ms = comm->last_ms; ctx->shard_tm.gather_bytes = comm->last_bytes;
    const uint64_t n_ret = is_root ? total : n_local;
    void* res = result_alloc(std::max<size_t>(1, (size_t)n_ret * rec));
    if (!res) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
    if (is_root) {
        // rank order on the device; own records first on the host
        size_t off_root = 0; for (int g = 0; g < root; ++g) off_root += bytes[g];
        hipError_t e = hipSuccess;
        size_t at = 0;
        if (bytes[root]) { e = hipMemcpyAsync(res, (const char*)d_all + off_root, bytes[root], hipMemcpyDeviceToHost, s); at += bytes[root]; }
        if (e == hipSuccess && off_root) { e = hipMemcpyAsync((char*)res + at, d_all, off_root, hipMemcpyDeviceToHost, s); at += off_root; }
        const size_t after = off_root + bytes[root], rest = (size_t)total * rec - after;
        if (e == hipSuccess && rest) e = hipMemcpyAsync((char*)res + at, (const char*)d_all + after, rest, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { necat_free(res); return set_err(ctx, NECAT_ERR_DEVICE, "record copy failed: %s", hipGetErrorString(e)); }
    } else if (n_local) {
        hipError_t e = hipMemcpyAsync(res, d_local, (size_t)n_local * rec, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { necat_free(res); return set_err(ctx, NECAT_ERR_DEVICE, "record copy failed: %s", hipGetErrorString(e)); }
    }
    *host_out = res; *n_out = n_ret;
    return NECAT_OK;
}
}  // namespace

int necat_find_candidates_sharded(necat_ctx* ctx, necat_comm* comm, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                                  int read_start_id, int ref_start_id, int pairwise, const necat_map_options* opt, int chunk_reads, int root,
                                  necat_candidate** out, uint64_t* n_out, uint64_t* n_local)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !comm || !ix || !ref || !reads || !opt || !out || !n_out || chunk_reads < 1 || root < 0 || root >= comm->nranks) return NECAT_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (n_local) *n_local = 0;
    ReadSel sel; sel.lo = comm->rank; sel.hi = comm->rank + 1; sel.nparts = comm->nranks; sel.chunk = chunk_reads;
    DevCands dev;
    int rc = find_impl(ctx, ix, ref, reads, read_start_id, ref_start_id, pairwise, opt, nullptr, nullptr, &dev, &sel);
    if (!rc && n_local) *n_local = dev.n;
    void* res = nullptr;
    if ((rc = gather_records(ctx, comm, root, dev.d, dev.n, sizeof(necat_candidate), &res, n_out, rc))) return rc;
    *out = (necat_candidate*)res;
    return NECAT_OK;
}

int necat_map_pair_sharded(necat_ctx* ctx, necat_comm* comm, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                           int read_start_id, int ref_start_id, int pairwise, const necat_map_options* opt, int tail_match_len, int chunk_reads, int root,
                           necat_m4** out, uint64_t* n_out, uint64_t* n_local, uint64_t* n_candidates)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !comm || !ix || !ref || !reads || !opt || !out || !n_out || chunk_reads < 1 || root < 0 || root >= comm->nranks) return NECAT_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (n_local) *n_local = 0;
    if (n_candidates) *n_candidates = 0;
    necat_map_options o = *opt;
    o.job = 1;
    ReadSel sel; sel.lo = comm->rank; sel.hi = comm->rank + 1; sel.nparts = comm->nranks; sel.chunk = chunk_reads;
    DevCands dev;
    int rc = find_impl(ctx, ix, ref, reads, read_start_id, ref_start_id, pairwise, &o, nullptr, nullptr, &dev, &sel);
    if (!rc && n_candidates) *n_candidates = dev.n;
    DevOut dout;
    ctx->tm.extend_ms = 0;
    if (!rc && dev.n) rc = extend_impl(ctx, ref, reads, read_start_id, ref_start_id, nullptr, dev.n, &o, tail_match_len, nullptr, nullptr, nullptr, &dev, &dout);
    if (!rc && n_local) *n_local = dout.n;
    void* res = nullptr;
    if ((rc = gather_records(ctx, comm, root, dout.d, dout.n, sizeof(necat_m4), &res, n_out, rc))) return rc;
    *out = (necat_m4*)res;
    return NECAT_OK;
}

// ---- a share of one (reference volume, query volume) pair: the building block of the pair scheduler (pair_sched.h).  No collective.
int necat_find_candidates_part(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                               int read_start_id, int ref_start_id, int pairwise, const necat_map_options* opt,
                               int chunk_reads, int slot_lo, int slot_hi, int slots, necat_candidate** out, uint64_t* n_out)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ix || !ref || !reads || !opt || !out || !n_out || chunk_reads < 1 || slots < 1 || slot_lo < 0 || slot_hi < slot_lo || slot_hi > slots) return NECAT_ERR_ARG;
    *out = nullptr; *n_out = 0;
    ReadSel sel; sel.lo = slot_lo; sel.hi = slot_hi; sel.nparts = slots; sel.chunk = chunk_reads; sel.always = true;
    return find_impl(ctx, ix, ref, reads, read_start_id, ref_start_id, pairwise, opt, out, n_out, nullptr, &sel);
}

int necat_map_pair_part(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                        int read_start_id, int ref_start_id, int pairwise, const necat_map_options* opt, int tail_match_len,
                        int chunk_reads, int slot_lo, int slot_hi, int slots, necat_m4** out, uint64_t* n_out, uint64_t* n_candidates)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ix || !ref || !reads || !opt || !out || !n_out || chunk_reads < 1 || slots < 1 || slot_lo < 0 || slot_hi < slot_lo || slot_hi > slots) return NECAT_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (n_candidates) *n_candidates = 0;
    necat_map_options o = *opt;
    o.job = 1;
    ReadSel sel; sel.lo = slot_lo; sel.hi = slot_hi; sel.nparts = slots; sel.chunk = chunk_reads; sel.always = true;
    DevCands dev;
    int rc = find_impl(ctx, ix, ref, reads, read_start_id, ref_start_id, pairwise, &o, nullptr, nullptr, &dev, &sel);
    if (rc) return rc;
    if (n_candidates) *n_candidates = dev.n;
    ctx->tm.extend_ms = 0;
    if (dev.n == 0) return NECAT_OK;
    return extend_impl(ctx, ref, reads, read_start_id, ref_start_id, nullptr, dev.n, &o, tail_match_len, out, n_out, nullptr, &dev);
}

int necat_pair_chunk_reads(uint64_t query_reads, int slots) { return slots < 1 ? NECAT_ERR_ARG : necat_host::pair_chunk_reads(query_reads, slots); }

int necat_pair_schedule(const uint64_t* vol_bases, int num_volumes, int nranks, int slots, necat_pair_unit** units, uint64_t** rank_off, int32_t** team)
{
    if (!vol_bases || num_volumes < 1 || nranks < 1 || slots < 1 || !units || !rank_off) return NECAT_ERR_ARG;
    const necat_host::PairSchedule S = necat_host::pair_schedule(vol_bases, num_volumes, nranks, slots);
    necat_pair_unit* u = (necat_pair_unit*)malloc(std::max<size_t>(1, S.units.size()) * sizeof(necat_pair_unit));
    uint64_t* ro = (uint64_t*)malloc(((size_t)nranks + 1) * 8);
    int32_t* tm = team ? (int32_t*)malloc((size_t)num_volumes * 8) : nullptr;
    if (!u || !ro || (team && !tm)) { free(u); free(ro); free(tm); return NECAT_ERR_MEMORY; }
    for (size_t i = 0; i < S.units.size(); ++i) { u[i].ref_vol = S.units[i].ref_vol; u[i].query_vol = S.units[i].query_vol; u[i].slot_lo = S.units[i].slot_lo; u[i].slot_hi = S.units[i].slot_hi; }
    for (int g = 0; g <= nranks; ++g) ro[g] = S.rank_off[(size_t)g];
    if (team) for (int v = 0; v < num_volumes; ++v) { tm[2 * v] = S.team_lo[(size_t)v]; tm[2 * v + 1] = S.team_hi[(size_t)v]; }
    *units = u; *rank_off = ro;
    if (team) *team = tm;
    return NECAT_OK;
}
