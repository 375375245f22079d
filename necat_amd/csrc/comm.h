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

// Every rank reports the status of the step it has just finished; all ranks leave with the same verdict.  A rank-local
// failure (an allocation, a kernel error flag, a capacity overflow) must not let this rank return while its peers walk into
// the next collective and wait for it forever: the sharded entry points call this before every exchange.
// Returns `rc` itself when this rank failed, NECAT_ERR_COMM when only another rank did, NECAT_OK when nobody did.
inline int agree(necat_ctx* ctx, necat_comm* c, int rc)
{
    if (c->nranks == 1) return rc;
    std::vector<int> all(c->nranks, 0);
    const int rg = c->gather(c->user, &rc, all.data(), sizeof(int));
    if (rc) return rc;
    if (rg) return set_err(ctx, NECAT_ERR_COMM, "host all-gather callback failed (%d)", rg);
    for (int r = 0; r < c->nranks; ++r)
        if (all[r]) return set_err(ctx, NECAT_ERR_COMM, "rank %d failed (status %d): the collective step is abandoned on every rank", r, all[r]);
    return NECAT_OK;
}

// first error of a sequence of calls that must ALL be issued (an open ncclGroup has to be closed, opened IPC handles
// closed, the barrier reached) whatever the earlier ones returned
struct FirstErr {
    necat_ctx* ctx; necat_comm* c; int rc = NECAT_OK;
    void hip(hipError_t e, const char* what) { if (e != hipSuccess && !rc) rc = set_err(ctx, NECAT_ERR_DEVICE, "%s failed: %s", what, hipGetErrorString(e)); }
    void nccl(ncclResult_t r, const char* what) { if (r != ncclSuccess && !rc) rc = set_err(ctx, NECAT_ERR_COMM, "%s failed: %s", what, c->p_GetErrorString ? c->p_GetErrorString(r) : "?"); }
    void keep(int r) { if (r && !rc) rc = r; }
};

// First contact (round 6): the first bytes a new RCCL communicator moves are a ring of small messages - rank r sends a pattern of its own to
// rank r + 1 and receives rank r - 1's - checked byte for byte, before any index slice or record depends on the transport.  `dbuf`: 2 x `bytes`
// of device memory (send half, receive half).  Every rank returns the same verdict (the ranks exchange their status through the host callback):
// NECAT_OK, or NECAT_ERR_COMM with the RCCL / HIP error text of THIS rank, or "rank %d failed".  necat_comm_create runs it when the ranks sit on
// distinct devices: a transport that cannot move 4 KB fails (or, with transport "auto", falls back to HIP IPC) at creation, with the reason.
inline unsigned char contact_byte(int rank, size_t i) { return (unsigned char)(i * 131u + 7u + 101u * (unsigned)rank); }
inline int first_contact(necat_ctx* ctx, necat_comm* c, void* dbuf, size_t bytes, hipStream_t s)
{
    if (c->nranks < 2 || c->transport != 0) return NECAT_OK;
    const int to = (c->rank + 1) % c->nranks, from = (c->rank - 1 + c->nranks) % c->nranks;
    std::vector<unsigned char> h(bytes), g(bytes, 0);
    for (size_t i = 0; i < bytes; ++i) h[i] = contact_byte(c->rank, i);
    unsigned char* d = (unsigned char*)dbuf;
    FirstErr fe{ctx, c};
    fe.hip(hipMemcpyAsync(d, h.data(), bytes, hipMemcpyHostToDevice, s), "hipMemcpyAsync");
    fe.hip(hipMemcpyAsync(d + bytes, g.data(), bytes, hipMemcpyHostToDevice, s), "hipMemcpyAsync");
    fe.nccl(c->p_GroupStart(), "ncclGroupStart");
    if (!fe.rc) {
        fe.nccl(c->p_Send(d, bytes, ncclChar, to, c->nccl, s), "ncclSend (first contact)");
        fe.nccl(c->p_Recv(d + bytes, bytes, ncclChar, from, c->nccl, s), "ncclRecv (first contact)");
    }
    fe.nccl(c->p_GroupEnd(), "ncclGroupEnd");            // the group is closed whatever was queued
    fe.hip(hipMemcpyAsync(g.data(), d + bytes, bytes, hipMemcpyDeviceToHost, s), "hipMemcpyAsync");
    fe.hip(hipStreamSynchronize(s), "hipStreamSynchronize");
    if (!fe.rc) for (size_t i = 0; i < bytes; ++i)
        if (g[i] != contact_byte(from, i)) { fe.keep(set_err(ctx, NECAT_ERR_COMM, "RCCL first contact: byte %zu of rank %d's message arrived as %u, not %u", i, from, (unsigned)g[i], (unsigned)contact_byte(from, i))); break; }
    return agree(ctx, c, fe.rc);
}

// One part of a distributed buffer: `bytes` bytes at `off` from the buffer's base on every rank.
struct Part { size_t off, bytes; };

// the (peer, direction) steps of an all-gather-v among nranks ranks as rank `rank` issues them: in step d it sends its own
// part to rank + d and receives part `from` from rank - d (staggered, so no peer is everybody's first target)
struct Xfer { int to, from; };
inline std::vector<Xfer> allgather_steps(int rank, int nranks)
{
    std::vector<Xfer> v;
    for (int d = 1; d < nranks; ++d) v.push_back(Xfer{(rank + d) % nranks, (rank - d + nranks) % nranks});
    return v;
}

// In-place all-gather-v: every rank holds its own part (parts[rank]) of `base` already in place and receives all others.
// The data of this rank must be complete on `s` (the exchange is ordered behind the stream's earlier work).
inline int allgatherv_inplace(necat_ctx* ctx, necat_comm* c, void* base, const std::vector<Part>& parts, hipStream_t s)
{
    const double t0 = now_ms();
    unsigned long long got = 0;
    char* b = (char*)base;
    FirstErr fe{ctx, c};
    if (c->nranks > 1 && c->transport == 0) {
        fe.nccl(c->p_GroupStart(), "ncclGroupStart");
        if (!fe.rc) for (const Xfer& x : allgather_steps(c->rank, c->nranks)) {
            if (parts[c->rank].bytes) fe.nccl(c->p_Send(b + parts[c->rank].off, parts[c->rank].bytes, ncclChar, x.to, c->nccl, s), "ncclSend");
            if (parts[x.from].bytes) { fe.nccl(c->p_Recv(b + parts[x.from].off, parts[x.from].bytes, ncclChar, x.from, c->nccl, s), "ncclRecv"); got += parts[x.from].bytes; }
        }
        fe.nccl(c->p_GroupEnd(), "ncclGroupEnd");            // the group is closed whatever was queued
        fe.hip(hipStreamSynchronize(s), "hipStreamSynchronize");
    } else if (c->nranks > 1) {
        // IPC pull: publish the handle of the allocation that holds `base`, open the peers', copy their parts
        void* abase = nullptr; size_t asize = 0;
        struct Msg { hipIpcMemHandle_t h; unsigned long long delta; int ok; } mine, *all;
        std::vector<Msg> msgs(c->nranks);
        all = msgs.data();
        memset(&mine, 0, sizeof mine);
        fe.hip(hipMemGetAddressRange((hipDeviceptr_t*)&abase, &asize, (hipDeviceptr_t)base), "hipMemGetAddressRange");
        if (!fe.rc) fe.hip(hipIpcGetMemHandle(&mine.h, abase), "hipIpcGetMemHandle");
        mine.delta = (unsigned long long)((char*)base - (char*)abase);
        fe.hip(hipStreamSynchronize(s), "hipStreamSynchronize");                   // my part is complete before anybody reads it
        mine.ok = fe.rc == NECAT_OK;
        fe.keep(host_allgather(ctx, c, &mine, all, sizeof(Msg)));                  // (doubles as the "data ready" barrier)
        bool all_ok = !fe.rc;
        for (int r = 0; r < c->nranks && all_ok; ++r) if (!all[r].ok) { all_ok = false; fe.keep(set_err(ctx, NECAT_ERR_COMM, "rank %d could not publish its buffer", r)); }
        std::vector<void*> opened(c->nranks, nullptr);
        if (all_ok) for (const Xfer& x : allgather_steps(c->rank, c->nranks)) {
            const int from = x.from;
            if (!parts[from].bytes || fe.rc) continue;
            fe.hip(hipIpcOpenMemHandle(&opened[from], all[from].h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
            if (fe.rc) { opened[from] = nullptr; continue; }
            const char* src = (const char*)opened[from] + all[from].delta + parts[from].off;
            fe.hip(hipMemcpyAsync(b + parts[from].off, src, parts[from].bytes, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync (peer)");
            got += parts[from].bytes;
        }
        fe.hip(hipStreamSynchronize(s), "hipStreamSynchronize");
        for (void* p : opened) if (p) fe.hip(hipIpcCloseMemHandle(p), "hipIpcCloseMemHandle");
        fe.keep(barrier(ctx, c));                                                   // nobody reuses its buffer while a peer still reads it
    }
    c->last_ms = now_ms() - t0; c->last_bytes = got;
    return fe.rc;
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
    FirstErr fe{ctx, c};
    if (c->rank == root && mine) fe.hip(hipMemcpyAsync((char*)recv + offs[root], send, mine, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
    if (c->nranks > 1 && c->transport == 0) {
        fe.nccl(c->p_GroupStart(), "ncclGroupStart");
        if (!fe.rc) {
            if (c->rank == root) {
                for (int r = 0; r < c->nranks; ++r)
                    if (r != root && counts[r]) { fe.nccl(c->p_Recv((char*)recv + offs[r], counts[r], ncclChar, r, c->nccl, s), "ncclRecv"); got += counts[r]; }
            } else if (mine) fe.nccl(c->p_Send(send, mine, ncclChar, root, c->nccl, s), "ncclSend");
        }
        fe.nccl(c->p_GroupEnd(), "ncclGroupEnd");
        fe.hip(hipStreamSynchronize(s), "hipStreamSynchronize");
    } else if (c->nranks > 1) {
        struct Msg { hipIpcMemHandle_t h; unsigned long long delta; int ok; } mine_m, *all;
        std::vector<Msg> msgs(c->nranks);
        all = msgs.data();
        memset(&mine_m, 0, sizeof mine_m);
        if (mine) {
            void* abase = nullptr; size_t asize = 0;
            fe.hip(hipMemGetAddressRange((hipDeviceptr_t*)&abase, &asize, (hipDeviceptr_t)send), "hipMemGetAddressRange");
            if (!fe.rc) fe.hip(hipIpcGetMemHandle(&mine_m.h, abase), "hipIpcGetMemHandle");
            mine_m.delta = (unsigned long long)((const char*)send - (const char*)abase);
        }
        fe.hip(hipStreamSynchronize(s), "hipStreamSynchronize");
        mine_m.ok = fe.rc == NECAT_OK;
        fe.keep(host_allgather(ctx, c, &mine_m, all, sizeof(Msg)));
        bool all_ok = !fe.rc;
        for (int r = 0; r < c->nranks && all_ok; ++r) if (!all[r].ok) { all_ok = false; fe.keep(set_err(ctx, NECAT_ERR_COMM, "rank %d could not publish its records", r)); }
        if (c->rank == root && all_ok) {
            std::vector<void*> opened(c->nranks, nullptr);
            for (int r = 0; r < c->nranks; ++r) {
                if (r == root || !counts[r] || fe.rc) continue;
                fe.hip(hipIpcOpenMemHandle(&opened[r], all[r].h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
                if (fe.rc) { opened[r] = nullptr; continue; }
                fe.hip(hipMemcpyAsync((char*)recv + offs[r], (const char*)opened[r] + all[r].delta, counts[r], hipMemcpyDeviceToDevice, s), "hipMemcpyAsync (peer)");
                got += counts[r];
            }
            fe.hip(hipStreamSynchronize(s), "hipStreamSynchronize");
            for (void* p : opened) if (p) fe.hip(hipIpcCloseMemHandle(p), "hipIpcCloseMemHandle");
        }
        fe.keep(barrier(ctx, c));
    } else fe.hip(hipStreamSynchronize(s), "hipStreamSynchronize");
    c->last_ms = now_ms() - t0; c->last_bytes = got;
    return fe.rc;
}

}  // namespace comm
}  // namespace necat
