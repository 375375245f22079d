// stage_volumes.inl - volumes: upload, pack, free.
// One of the stage files of libnecat_hip.so's single translation unit: necat_hip.hip includes them in order, inside its extern "C" block, after the
// context / knob / result-pool code they all use (the kernels are header templates and the stages share host helpers: one device code object, one 30 s build).

// ------------------------------------------------------------------------------------------ volumes

int necat_volume_upload(necat_ctx* ctx, const uint8_t* pac, uint64_t nbases, const uint64_t* seq_offset,
                        const uint64_t* seq_size, uint64_t nseq, necat_volume** out)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !out || (nbases && !pac) || (nseq && (!seq_offset || !seq_size))) return NECAT_ERR_ARG;
    *out = nullptr;
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    // the overlap stage requires the reads of a volume to tile it in order (packed_db.c:229-253)
    uint64_t run = 0;
    for (uint64_t i = 0; i < nseq; ++i) {
        if (seq_offset[i] != run) return set_err(ctx, NECAT_ERR_ARG, "sequence %lu does not start where sequence %lu ends", (unsigned long)i, (unsigned long)(i - 1));
        run += seq_size[i];
    }
    if (run != nbases) return set_err(ctx, NECAT_ERR_ARG, "sequence sizes sum to %lu, volume holds %lu bases", (unsigned long)run, (unsigned long)nbases);
    if (nbases >= (1ULL << 32)) return set_err(ctx, NECAT_ERR_ARG, "volume too large (>= 2^32 bases; oc2mkdb cuts volumes at 2e9, makedb/main.c:8)");
    necat_volume* v = new necat_volume();
    v->nbases = nbases; v->nseq = nseq;
    uint64_t* staging = nullptr;
    // everything allocated so far goes when a step fails (a long-lived context must not leak device memory on an error)
    auto upload = [&]() -> int {
        const uint64_t nwords = (nbases + 31) / 32;
        const uint64_t pac_bytes = (nbases + 3) / 4;
        NECAT_HIP(ctx, hipMalloc((void**)&v->bases_alloc, (nwords + 2 * kGuardWords) * 8));
        NECAT_HIP(ctx, hipMemsetAsync(v->bases_alloc, 0, (nwords + 2 * kGuardWords) * 8, ctx->stream));
        v->bases = v->bases_alloc + kGuardWords;
        if (nwords) {
            NECAT_HIP(ctx, hipMalloc((void**)&staging, nwords * 8));
            NECAT_HIP(ctx, hipMemsetAsync(staging, 0, nwords * 8, ctx->stream));
            NECAT_HIP(ctx, hipMemcpyAsync(staging, pac, pac_bytes, hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_repack, dim3(grid_for(nwords, 256, 65536)), dim3(256), 0, ctx->stream, staging, nwords, v->bases);
            NECAT_CHECK_LAUNCH(ctx, "k_repack");
            NECAT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        v->h_seq_off.resize(nseq + 1);
        for (uint64_t i = 0; i < nseq; ++i) v->h_seq_off[i] = seq_offset[i];
        v->h_seq_off[nseq] = nbases;
        NECAT_HIP(ctx, hipMalloc((void**)&v->seq_off, (nseq + 1) * 8));
        NECAT_HIP(ctx, hipMemcpyAsync(v->seq_off, v->h_seq_off.data(), (nseq + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        NECAT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return NECAT_OK;
    };
    const int rc = upload();
    if (staging) (void)hipFree(staging);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); necat_volume_free(ctx, v); return rc; }
    *out = v;
    return NECAT_OK;
}

// oc2mkdb's packing step on the device: ASCII bases -> pac bytes (what the volume file holds) and, when the
// caller wants it, the resident device volume in the same go.
int necat_volume_pack(necat_ctx* ctx, const char* ascii, uint64_t nbases, const uint64_t* seq_offset, const uint64_t* seq_size,
                      uint64_t nseq, uint8_t* pac_out, necat_volume** out)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || (nbases && !ascii) || (!pac_out && !out)) return NECAT_ERR_ARG;
    if (out) *out = nullptr;
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t pac_bytes = (nbases + 3) / 4;
    std::vector<uint8_t> own;
    uint8_t* pac = pac_out;
    if (!pac) { own.resize(pac_bytes + 8); pac = own.data(); }
    // pieces of <= 256 M bases: 256 MB of text + 64 MB of pac on the device at a time
    const uint64_t piece = 1ULL << 28;
    unsigned char *d_txt = nullptr, *d_pac = nullptr;
    auto body = [&]() -> int {
        if (!nbases) return NECAT_OK;
        const uint64_t cap = std::min(piece, nbases);
        NECAT_HIP(ctx, hipMalloc((void**)&d_txt, cap)); NECAT_HIP(ctx, hipMalloc((void**)&d_pac, cap / 4 + 8));
        for (uint64_t b0 = 0; b0 < nbases; b0 += piece) {
            const uint64_t nb = std::min(piece, nbases - b0);
            NECAT_HIP(ctx, hipMemcpyAsync(d_txt, ascii + b0, nb, hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_pack_ascii, dim3(grid_for((nb + 3) / 4, 256, 1u << 16)), dim3(256), 0, ctx->stream, d_txt, nb, b0, d_pac);
            NECAT_CHECK_LAUNCH(ctx, "k_pack_ascii");
            NECAT_HIP(ctx, hipMemcpyAsync(pac + b0 / 4, d_pac, (nb + 3) / 4, hipMemcpyDeviceToHost, ctx->stream));
            NECAT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        return NECAT_OK;
    };
    int rc = body();
    if (d_txt) (void)hipFree(d_txt);
    if (d_pac) (void)hipFree(d_pac);
    if (rc) return rc;
    if (out) rc = necat_volume_upload(ctx, pac, nbases, seq_offset, seq_size, nseq, out);
    return rc;
}

void necat_volume_free(necat_ctx* ctx, necat_volume* v)
{
    KnobScope knob_scope_(ctx);
    if (!v) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (v->bases_alloc) (void)hipFree(v->bases_alloc);
    if (v->seq_off) (void)hipFree(v->seq_off);
    delete v;
}
