// comm.h - the rank-to-rank data path of a single-volume multi-GPU job (fine granularity): one process per
// GPU, the k-mer index built in hash-range slices and all-gathered, query reads dealt out in chunks, records gathered
// on a root rank.
//
// Two transports move device memory between ranks:
//   RCCL   ncclSend / ncclRecv inside one ncclGroup per exchange - the direct all-pairs pattern: on MI355X every GPU
//          has its own xGMI link to each of its 7 peers (~153 GB/s per link), so all slices travel concurrently, one per
//          link; a ring would push everything through one link per GPU.  librccl is opened at run time (dlopen), so the
//          single-GPU library has no RCCL dependency; inside a process that already loaded RCCL (torch.distributed) the
//          same copy is used.
//   IPC    hipIpcMemHandle + device-to-device copies, for ranks that share a device (RCCL refuses two ranks on one GPU):
//          this is what lets the whole multi-rank path be parity-tested on a 1-GPU box.  Same pull pattern, same bytes.
// Small host-side exchanges (list sizes, the ncclUniqueId, IPC handles, barriers) go through a caller-supplied all-gather
// callback: bench.py backs it with torch.distributed, the oc2pmov launcher with pipes between its worker processes.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "runtime.h"

struct necat_comm {
    int rank = 0, nranks = 1;
    necat_host_allgather_fn gather = nullptr;
    void* user = nullptr;
    int transport = 0;               // 0 = RCCL, 1 = HIP IPC
    // ---- RCCL
    void* lib = nullptr;
    ncclComm_t nccl = nullptr;
    ncclResult_t (*p_GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*p_CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*p_CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*p_Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*p_Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*p_GroupStart)() = nullptr;
    ncclResult_t (*p_GroupEnd)() = nullptr;
    const char* (*p_GetErrorString)(ncclResult_t) = nullptr;
    // ---- accounting of the last exchanges (ms of wall time on this rank, bytes received)
    double last_ms = 0; unsigned long long last_bytes = 0;
};

namespace necat {
namespace comm {

inline double now_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

inline int host_allgather(necat_ctx* ctx, necat_comm* c, const void* send, void* recv, size_t bytes)
{
    if (c->nranks == 1) { memcpy(recv, send, bytes); return NECAT_OK; }
    const int rc = c->gather(c->user, send, recv, bytes);
    if (rc) return set_err(ctx, NECAT_ERR_COMM, "host all-gather callback failed (%d)", rc);
    return NECAT_OK;
}

inline int barrier(necat_ctx* ctx, necat_comm* c)
{
    std::vector<int> v(c->nranks);
    const int me = c->rank;
    return host_allgather(ctx, c, &me, v.data(), sizeof(int));
}

#define NECAT_NCCL(ctx, c, call) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) \
    return necat::set_err(ctx, NECAT_ERR_COMM, "%s failed: %s", #call, (c)->p_GetErrorString ? (c)->p_GetErrorString(r__) : "?"); } while (0)

inline int load_rccl(necat_ctx* ctx, necat_comm* c)
{
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { c->lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (c->lib) break; }
    if (!c->lib) return set_err(ctx, NECAT_ERR_COMM, "cannot open librccl: %s", dlerror());
#define NECAT_SYM(field, name) do { *(void**)&c->field = dlsym(c->lib, name); \
    if (!c->field) return set_err(ctx, NECAT_ERR_COMM, "librccl lacks %s", name); } while (0)
    NECAT_SYM(p_GetUniqueId, "ncclGetUniqueId"); NECAT_SYM(p_CommInitRank, "ncclCommInitRank"); NECAT_SYM(p_CommDestroy, "ncclCommDestroy");
    NECAT_SYM(p_Send, "ncclSend"); NECAT_SYM(p_Recv, "ncclRecv"); NECAT_SYM(p_GroupStart, "ncclGroupStart"); NECAT_SYM(p_GroupEnd, "ncclGroupEnd");
    NECAT_SYM(p_GetErrorString, "ncclGetErrorString");
#undef NECAT_SYM
    return NECAT_OK;
}

// One part of a distributed buffer: `bytes` bytes at `off` from the buffer's base on every rank.
struct Part { size_t off, bytes; };

// In-place all-gather-v: every rank holds its own part (parts[rank]) of `base` already in place and receives all others.
// The data of this rank must be complete on `s` (the exchange is ordered behind the stream's earlier work).
inline int allgatherv_inplace(necat_ctx* ctx, necat_comm* c, void* base, const std::vector<Part>& parts, hipStream_t s)
{
    const double t0 = now_ms();
    unsigned long long got = 0;
    char* b = (char*)base;
    if (c->nranks > 1 && c->transport == 0) {
        NECAT_NCCL(ctx, c, c->p_GroupStart());
        for (int d = 1; d < c->nranks; ++d) {
            // staggered peer order: in step d every rank sends to rank + d and receives from rank - d, so no peer is everybody's first target
            const int to = (c->rank + d) % c->nranks, from = (c->rank - d + c->nranks) % c->nranks;
            if (parts[c->rank].bytes) NECAT_NCCL(ctx, c, c->p_Send(b + parts[c->rank].off, parts[c->rank].bytes, ncclChar, to, c->nccl, s));
            if (parts[from].bytes) { NECAT_NCCL(ctx, c, c->p_Recv(b + parts[from].off, parts[from].bytes, ncclChar, from, c->nccl, s)); got += parts[from].bytes; }
        }
        NECAT_NCCL(ctx, c, c->p_GroupEnd());
        NECAT_HIP(ctx, hipStreamSynchronize(s));
    } else if (c->nranks > 1) {
        // IPC pull: publish the handle of the allocation that holds `base`, open the peers', copy their parts
        void* abase = nullptr; size_t asize = 0;
        NECAT_HIP(ctx, hipMemGetAddressRange((hipDeviceptr_t*)&abase, &asize, (hipDeviceptr_t)base));
        struct Msg { hipIpcMemHandle_t h; unsigned long long delta; } mine, *all;
        std::vector<Msg> msgs(c->nranks);
        all = msgs.data();
        NECAT_HIP(ctx, hipIpcGetMemHandle(&mine.h, abase));
        mine.delta = (unsigned long long)((char*)base - (char*)abase);
        NECAT_HIP(ctx, hipStreamSynchronize(s));                                   // my part is complete before anybody reads it
        int rc = host_allgather(ctx, c, &mine, all, sizeof(Msg));                  // (doubles as the "data ready" barrier)
        if (rc) return rc;
        std::vector<void*> opened(c->nranks, nullptr);
        for (int d = 1; d < c->nranks; ++d) {
            const int from = (c->rank - d + c->nranks) % c->nranks;
            if (!parts[from].bytes) continue;
            NECAT_HIP(ctx, hipIpcOpenMemHandle(&opened[from], all[from].h, hipIpcMemLazyEnablePeerAccess));
            const char* src = (const char*)opened[from] + all[from].delta + parts[from].off;
            NECAT_HIP(ctx, hipMemcpyAsync(b + parts[from].off, src, parts[from].bytes, hipMemcpyDeviceToDevice, s));
            got += parts[from].bytes;
        }
        NECAT_HIP(ctx, hipStreamSynchronize(s));
        for (void* p : opened) if (p) NECAT_HIP(ctx, hipIpcCloseMemHandle(p));
        if ((rc = barrier(ctx, c))) return rc;                                      // nobody reuses its buffer while a peer still reads it
    }
    c->last_ms = now_ms() - t0; c->last_bytes = got;
    return NECAT_OK;
}

// Gather-v to `root`: every rank contributes `bytes` bytes at `send` (device memory); the root receives rank r's
// contribution at recv + offs[r] (its own by a device copy).  counts[] = every rank's byte count (all ranks know them).
inline int gatherv(necat_ctx* ctx, necat_comm* c, const void* send, const std::vector<size_t>& counts, int root, void* recv, hipStream_t s)
{
    const double t0 = now_ms();
    unsigned long long got = 0;
    std::vector<size_t> offs(c->nranks + 1, 0);
    for (int r = 0; r < c->nranks; ++r) offs[r + 1] = offs[r] + counts[r];
    const size_t mine = counts[c->rank];
    if (c->rank == root && mine) NECAT_HIP(ctx, hipMemcpyAsync((char*)recv + offs[root], send, mine, hipMemcpyDeviceToDevice, s));
    if (c->nranks > 1 && c->transport == 0) {
        NECAT_NCCL(ctx, c, c->p_GroupStart());
        if (c->rank == root) {
            for (int r = 0; r < c->nranks; ++r)
                if (r != root && counts[r]) { NECAT_NCCL(ctx, c, c->p_Recv((char*)recv + offs[r], counts[r], ncclChar, r, c->nccl, s)); got += counts[r]; }
        } else if (mine) NECAT_NCCL(ctx, c, c->p_Send(send, mine, ncclChar, root, c->nccl, s));
        NECAT_NCCL(ctx, c, c->p_GroupEnd());
        NECAT_HIP(ctx, hipStreamSynchronize(s));
    } else if (c->nranks > 1) {
        struct Msg { hipIpcMemHandle_t h; unsigned long long delta; } mine_m, *all;
        std::vector<Msg> msgs(c->nranks);
        all = msgs.data();
        memset(&mine_m, 0, sizeof mine_m);
        if (mine) {
            void* abase = nullptr; size_t asize = 0;
            NECAT_HIP(ctx, hipMemGetAddressRange((hipDeviceptr_t*)&abase, &asize, (hipDeviceptr_t)send));
            NECAT_HIP(ctx, hipIpcGetMemHandle(&mine_m.h, abase));
            mine_m.delta = (unsigned long long)((const char*)send - (const char*)abase);
        }
        NECAT_HIP(ctx, hipStreamSynchronize(s));
        int rc = host_allgather(ctx, c, &mine_m, all, sizeof(Msg));
        if (rc) return rc;
        if (c->rank == root) {
            std::vector<void*> opened(c->nranks, nullptr);
            for (int r = 0; r < c->nranks; ++r) {
                if (r == root || !counts[r]) continue;
                NECAT_HIP(ctx, hipIpcOpenMemHandle(&opened[r], all[r].h, hipIpcMemLazyEnablePeerAccess));
                NECAT_HIP(ctx, hipMemcpyAsync((char*)recv + offs[r], (const char*)opened[r] + all[r].delta, counts[r], hipMemcpyDeviceToDevice, s));
                got += counts[r];
            }
            NECAT_HIP(ctx, hipStreamSynchronize(s));
            for (void* p : opened) if (p) NECAT_HIP(ctx, hipIpcCloseMemHandle(p));
        }
        if ((rc = barrier(ctx, c))) return rc;
    } else NECAT_HIP(ctx, hipStreamSynchronize(s));
    c->last_ms = now_ms() - t0; c->last_bytes = got;
    return NECAT_OK;
}

}  // namespace comm
}  // namespace necat
