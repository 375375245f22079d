// stage_pcan.inl - candidate partitions (oc2pcan).
// One of the stage files of libnecat_hip.so's single translation unit: necat_hip.hip includes them in order, inside its extern "C" block, after the
// context / knob / result-pool code they all use (the kernels are header templates and the stages share host helpers: one device code object, one 30 s build).

// ------------------------------------------------------------------------------------------ candidate partitions (oc2pcan)

int necat_pcan_partition(necat_ctx* ctx, const necat_candidate* cands, uint64_t n, int batch_size, int num_reads,
                         uint32_t** records, uint64_t** part_off, int* num_parts)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || (n && !cands) || !records || !part_off || !num_parts || batch_size < 1 || num_reads < 0) return NECAT_ERR_ARG;
    *records = nullptr; *part_off = nullptr;
    const int nparts = (int)(((int64_t)num_reads + batch_size - 1) / batch_size);       // pcan.c:111
    *num_parts = nparts;
    uint64_t* off = (uint64_t*)result_alloc((size_t)(nparts + 1) * 8);
    if (!off) return set_err(ctx, NECAT_ERR_MEMORY, "host allocation");
    for (int p = 0; p <= nparts; ++p) off[p] = 0;
    *part_off = off;
    if (n == 0 || nparts == 0) { *records = (uint32_t*)result_alloc(28); return NECAT_OK; }
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int rc;
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_EXT_CAND], n * sizeof(necat_candidate) + 2 * n * sizeof(PackedCan) + (size_t)nparts * 8 + 256))) return rc;
    char* b = (char*)ctx->scratch[SC_EXT_CAND].p;
    necat_candidate* d_c = (necat_candidate*)b; b += n * sizeof(necat_candidate);
    PackedCan* d_out = (PackedCan*)b; b += 2 * n * sizeof(PackedCan);
    unsigned long long* d_cur = (unsigned long long*)(((uintptr_t)b + 63) & ~(uintptr_t)63);
    NECAT_HIP(ctx, hipMemcpyAsync(d_c, cands, n * sizeof(necat_candidate), hipMemcpyHostToDevice, s));
    NECAT_HIP(ctx, hipMemsetAsync(d_cur, 0, (size_t)nparts * 8, s));
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_pcan<0>, dim3(grid), dim3(256), 0, s, (const necat_candidate*)d_c, n, batch_size, nparts, d_cur, (PackedCan*)nullptr);
    NECAT_CHECK_LAUNCH(ctx, "k_pcan<count>");
    std::vector<unsigned long long> cnt(nparts);
    NECAT_HIP(ctx, hipMemcpyAsync(cnt.data(), d_cur, (size_t)nparts * 8, hipMemcpyDeviceToHost, s));
    NECAT_HIP(ctx, hipStreamSynchronize(s));
    std::vector<unsigned long long> start(nparts);
    for (int p = 0; p < nparts; ++p) { start[p] = off[p]; off[p + 1] = off[p] + cnt[p]; }
    const uint64_t total = off[nparts];
    NECAT_HIP(ctx, hipMemcpyAsync(d_cur, start.data(), (size_t)nparts * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pcan<1>, dim3(grid), dim3(256), 0, s, (const necat_candidate*)d_c, n, batch_size, nparts, d_cur, d_out);
    NECAT_CHECK_LAUNCH(ctx, "k_pcan<scatter>");
    uint32_t* rec = (uint32_t*)result_alloc((size_t)total * 28 + 28);
    if (!rec) return set_err(ctx, NECAT_ERR_MEMORY, "host allocation");
    if (total) NECAT_HIP(ctx, hipMemcpyAsync(rec, d_out, (size_t)total * 28, hipMemcpyDeviceToHost, s));
    NECAT_HIP(ctx, hipStreamSynchronize(s));
    *records = rec;
    return NECAT_OK;
}
