// runtime.h - host-side plumbing of libnecat_hip.so: context, error capture, grow-only device
// buffers, HIP event timers.  No torch types, no CPU fallback: every failure is reported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <initializer_list>
#include <string>
#include <utility>
#include <vector>

#include "../../include/necat_hip.h"
#include "knobs.h"
#include "dev_common.h"

namespace necat {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

}  // namespace necat

constexpr int kNumEvents = 48;
constexpr unsigned kRoundRing = 1024;  // entries of necat_ctx::round_ring per lane (the ring holds kMaxExtLanes lanes' worth)
constexpr int kMaxExtLanes = 4;        // lanes of the extension rounds: the context's own set + up to three ExtLane1 (NECAT_EXT_LANES, default 2)

// The second lane of the extension rounds (stage_extend.inl, ExtLane): while one batch of candidates is in its last, latency-bound rounds
// the next batch runs its first, chip-filling ones beside it - on buffers, streams, events and a ring half of its own.  Lane 0 is the
// context's own set (scratch[SC_EXT_*], stream_a .. stream_d, ev[]); this is lane 1, created the first time two batches overlap.
struct ExtLane1 {
    necat::DevBuf buf[16];             // by role: ExtLaneBuf in stage_extend.inl
    hipStream_t st[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[kNumEvents] = {};    // (null = not created: ext_lane creates the missing ones, necat_ctx_destroy destroys the others)
    unsigned long long round_seq = 0;
    bool ready = false;
};

struct necat_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_a = nullptr, stream_b = nullptr;   // extension rounds: list A / list B run concurrently
    hipStream_t stream_c = nullptr;                       // second list-B stream (small lists alternate)
    hipStream_t stream_d = nullptr;                       // list A's ragged / wide blocks of a round whose full blocks run through ext_rcwalk.h
    bool serial_streams = false;                          // NECAT_SERIAL=1: stream_a .. stream_d are aliases of `stream`
    hipStream_t stream_copy = nullptr;                    // deferred device-to-host copies (alignment columns of the consensus loop)
    bool copy_pending = false;                            // a copy on stream_copy still reads SC_EXT_COLS_OUT (ev[17] marks its end)
    char err[1024] = {0};
    necat_timings tm;
    necat_shard_timings shard_tm;      // last sharded calls (necat_index_build_sharded, necat_*_sharded)
    hipEvent_t ev[kNumEvents];
    void* pin_plan = nullptr; size_t pin_plan_cap = 0;   // pinned host scratch of the seeding plan (hit counts down, order + scratch layout up)
    void* round_ring = nullptr;        // pinned, device-visible ring of RoundPub entries: list sizes published by the round kernels
    void* round_ring_dev = nullptr;    // the same memory as the device addresses it
    unsigned long long round_seq = 0;  // rounds published so far (the next round publishes round_seq + 1)
    necat::DevBuf scratch[64];         // grow-only arenas, indexed by purpose (ScratchId; SC_COUNT <= 64)
    uint64_t scratch_live = 0;         // bit i: a call in progress holds pointers into scratch[i] (necat::ArenaUse) - buf_ensure_lend never hands such an arena to another phase
    void* seed_ht_ptr = nullptr;       // the seeding hash arena (SC_SEED_HT) whose first seed_ht_clean bytes are known to be all-empty (0xFF):
    size_t seed_ht_cap = 0;            // .. and its capacity when that was established (a reallocation at the same address is a new arena)
    size_t seed_ht_clean = 0;          // every call leaves the arena as it found it (k_seed_clear resets the slots it used), so it is filled once per allocation
    char devname[256] = {0};
    int num_cu = 0;
    uint32_t epoch = 0;
    void* cns_scratch = nullptr;       // host buffers of the consensus loop kept between calls (necat::cns::Scratch)
    necat::Knobs knobs;                // this context's tuning / test knobs (knobs.h: read from the environment in necat_ctx_create)
    necat::DevBuf idx_cache[2];        // released index arrays kept for the next build (8.6 GB hipMalloc/hipFree per step otherwise)
    ExtLane1 lanex[kMaxExtLanes - 1];  // lanes 1 .. of the extension rounds (each created on first use)
};

struct necat_volume {
    uint64_t nbases = 0, nseq = 0;
    uint64_t* bases_alloc = nullptr;   // guarded allocation
    uint64_t* bases = nullptr;         // bases_alloc + guard
    uint64_t* seq_off = nullptr;       // [nseq + 1]
    std::vector<uint64_t> h_seq_off;   // host copy (offsets are tiny; used for planning)
};

struct necat_index {
    int k = 0;
    uint64_t table_entries = 0, n_offsets = 0;
    // the table: ONE allocation (`table`, stats_cap bytes) holding either the dense reference layout (kmer_stats) or the sparse
    // one (words: table_entries / 64 IdxWords, then n_compact non-zero entries) - necat::IndexView in dev_common.h
    void* table = nullptr;
    uint64_t* kmer_stats = nullptr;
    void* words = nullptr;
    uint64_t* compact = nullptr;
    uint64_t n_compact = 0;
    uint64_t* offset_list = nullptr;
    size_t stats_cap = 0, offs_cap = 0;   // allocation sizes in bytes
};

namespace necat {

// an entry point of the C ABI makes its context's knobs the current ones of this thread for the length of the call (knobs.h)
struct KnobScope {
    const Knobs* prev;
    explicit KnobScope(const necat_ctx* c) : prev(tl_knobs) { if (c) tl_knobs = &c->knobs; }
    ~KnobScope() { tl_knobs = prev; }
    KnobScope(const KnobScope&) = delete; KnobScope& operator=(const KnobScope&) = delete;
};

inline int set_err(necat_ctx* ctx, int code, const char* fmt, ...)
{
    if (ctx) {
        va_list ap; va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

#define NECAT_HIP(ctx, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) \
    return necat::set_err(ctx, NECAT_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); } while (0)

#define NECAT_CHECK_LAUNCH(ctx, name) do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) \
    return necat::set_err(ctx, NECAT_ERR_DEVICE, "launch of %s failed: %s", name, hipGetErrorString(e__)); } while (0)

constexpr int kGuardWords = 4;   // 128 bases of slack on both sides of a volume's bases

inline int buf_ensure(necat_ctx* ctx, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return NECAT_OK;
    if (b.p) { hipError_t e = hipFree(b.p); (void)e; b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr; b.cap = 0;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = tot = 0;
        char own[512]; int at = 0; size_t sum = 0;          // this context's arenas of 256 MB and more (ScratchId : MB)
        own[0] = 0;
        if (ctx) for (int i = 0; i < (int)(sizeof ctx->scratch / sizeof ctx->scratch[0]); ++i) {
            sum += ctx->scratch[i].cap;
            if (ctx->scratch[i].cap >= ((size_t)256 << 20) && at < 480) at += snprintf(own + at, sizeof own - (size_t)at, " %d:%zu", i, ctx->scratch[i].cap >> 20);
        }
        return set_err(ctx, NECAT_ERR_MEMORY, "hipMalloc(%zu) failed: %s (device memory: %zu MB free of %zu MB; this context's arenas: %zu MB, the big ones [id:MB]%s)", want,
                       hipGetErrorString(e), fr >> 20, tot >> 20, sum >> 20, own);
    }
    b.cap = want;
    return NECAT_OK;
}

// grow keeping the first `keep` bytes (device-to-device copy; capacity doubles)
inline int buf_grow(necat_ctx* ctx, DevBuf& b, size_t bytes, size_t keep, hipStream_t s)
{
    if (bytes <= b.cap) return NECAT_OK;
    size_t want = bytes > 2 * b.cap ? bytes + bytes / 8 + 256 : 2 * b.cap;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, want);
    if (e != hipSuccess) return set_err(ctx, NECAT_ERR_MEMORY, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    if (b.p && keep) {
        e = hipMemcpyAsync(q, b.p, keep, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { (void)hipFree(q); return set_err(ctx, NECAT_ERR_DEVICE, "device copy failed: %s", hipGetErrorString(e)); }
    }
    if (b.p) (void)hipFree(b.p);
    b.p = q; b.cap = want;
    return NECAT_OK;
}

enum ScratchId {
    SC_CNT32 = 0, SC_PARTIAL, SC_TMPLIST, SC_MISC,
    SC_SEED_META, SC_SEED_HT, SC_SEED_POOL, SC_SEED_CHAIN, SC_SEED_OUT, SC_SEED_FINAL,
    SC_EXT_TASKS, SC_EXT_LISTS, SC_EXT_FRAG, SC_EXT_MAT, SC_EXT_OPS, SC_EXT_RES, SC_EXT_CAND, SC_SMALL, SC_PART,
    SC_EXT_COLS, SC_EXT_COLS_OUT, SC_PART2, SC_SEED_ALL, SC_EXT_MATB, SC_EXT_MATB2, SC_EXT_PERM, SC_GATHER, SC_SPLIT, SC_SPLIT2,
    SC_ASM_BAND, SC_ASM_OPS, SC_ASM_COLS, SC_ASM_MISC, SC_ASM_FRAG, SC_ASM_OUT, SC_SEED_KST, SC_EXT_CKPT, SC_EXT_WOUT, SC_EXT_CKPTB, SC_EXT_CKPTB2, SC_EXT_WOUTB, SC_EXT_WOUTB2,
    SC_ASM_OCC, SC_ASM_TAB, SC_ASM_VMETA, SC_ASM_VHT, SC_ASM_VPOOL, SC_ASM_VOUT, SC_ASM_SEL, SC_ASM_RIDX, SC_ASM_RNEXT, SC_ASM_PAIRS, SC_ASM_SEEDS,
    SC_ASM_VMETA2, SC_ASM_VHT2, SC_ASM_VPOOL2, SC_ASM_VOUT2, SC_ASM_SEL2, SC_ASM_RIDX2, SC_ASM_RNEXT2,
    SC_STATS,
    SC_COUNT
};

// buf_ensure for arenas of DIFFERENT phases of a context (the index build's split buffers / the seeding arenas: never live at the same
// time, every call that uses them has finished with them when it returns): before allocating, take the buffer of a donor that is big
// enough and leave it this one's - the phases then hand ONE allocation back and forth instead of holding two (a 2 Gbp volume: 17 GB of
// split records and 14 GB of seed blocks; a process waits 30 - 55 ms per GB for device memory it maps the first time).
// What keeps that safe is no longer only the order of the donor lists: a phase marks the arenas it holds pointers into (ArenaUse, for the length of
// the call) and a marked arena is never taken - an index build running beside a candidate search of the same context would allocate instead of aliasing.
struct ArenaUse {
    necat_ctx* ctx; uint64_t mine;
    ArenaUse(necat_ctx* c, std::initializer_list<int> ids) : ctx(c), mine(0) { for (int i : ids) mine |= 1ULL << i; mine &= ~c->scratch_live; c->scratch_live |= mine; }
    ~ArenaUse() { ctx->scratch_live &= ~mine; }
    ArenaUse(const ArenaUse&) = delete; ArenaUse& operator=(const ArenaUse&) = delete;
};
inline int buf_ensure_lend(necat_ctx* ctx, int id, size_t bytes, std::initializer_list<int> donors)
{
    DevBuf& b = ctx->scratch[id];
    if (bytes <= b.cap) return NECAT_OK;
    static const bool off = getenv("NECAT_NO_LEND") && atoi(getenv("NECAT_NO_LEND"));
    if (!off) for (int d : donors) {
        DevBuf& o = ctx->scratch[d];
        if (d != id && o.cap >= bytes && !((ctx->scratch_live >> d) & 1ULL)) { std::swap(b, o); return NECAT_OK; }
    }
    return buf_ensure(ctx, b, bytes);
}

}  // namespace necat
