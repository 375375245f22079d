// necat_hip.hip - libnecat_hip.so: C ABI (include/necat_hip.h) + host orchestration of the gfx950
// kernels.  Written for MI355X only (wave64, 256 CUs, 288 GB HBM3E): buffers are sized for HBM
// residency of whole volumes, work lists and the traceback band of 10^5 concurrent alignments.
#include <algorithm>
#include <chrono>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <time.h>
#include <unistd.h>
#include <numeric>

#include "runtime.h"
#include "index_kernels.h"
#include "seed_kernels.h"
#include "pcan_kernels.h"
#include "ext_kernels.h"
#include "ext_tail.h"
#include "ext_rcwalk.h"
#include "ext_rcwalk3.h"
#include "asm_kernels.h"
#include "asm_coop.h"
#include "asm_plan.h"
#include "cns_loop.h"
#include "cns_rescue.h"
#include "rm_host.h"
#include "comm.h"
#include "pair_sched.h"

// The two lanes of the extension rounds are eight streams; the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of streams
// that share a queue run one after the other - with 4 queues the second lane gains 3.7 % at yeast size, with 8 it gains 7.7 % (tools/r05/run20.sh).  The variable is read
// when the runtime initialises (the first HIP call of the process), so it is set - unless the user has - when this library is loaded.
__attribute__((constructor)) static void necat_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

namespace necat { thread_local const Knobs* tl_knobs = nullptr; }      // knobs.h: set by KnobScope in every entry point that takes a context
using namespace necat;
static_assert(SC_COUNT <= (int)(sizeof(necat_ctx::scratch) / sizeof(necat::DevBuf)), "a ScratchId without an arena");

namespace {

// (re)allocate a band-record pool; a fresh allocation is zeroed once (not needed for correctness - the walk only reads
// records its round stored - but it keeps a run reproducible should that invariant ever break)
int ensure_zeroed(necat_ctx* ctx, DevBuf& b, size_t bytes, hipStream_t s)
{
    const void* before = b.p; const size_t cap0 = b.cap;
    int rc = buf_ensure(ctx, b, bytes);
    if (rc) return rc;
    if (b.p != before || b.cap != cap0) { hipError_t e = hipMemsetAsync(b.p, 0, b.cap, s); if (e != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "memset failed"); }
    return NECAT_OK;
}

inline unsigned grid_for(uint64_t n, unsigned block, unsigned cap = 1u << 20)
{
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

DevVolume dev_view(const necat_volume* v)
{
    DevVolume d; d.bases = v->bases; d.seq_off = v->seq_off; d.nbases = v->nbases; d.nseq = v->nseq;
    return d;
}

// (the knobs: knobs.h - per context since round 5)

// the recompute walk of a list of `nitems` work indices: one workgroup per 64 blocks (two waves: k_rcwalk3; four: k_rcwalk2w) or one wave per 16 (k_rcwalk2)
template <int NW, int TW, int COLS, int MAXOPS, class... A>
static void launch_rcwalk2(u32 nitems, hipStream_t s, A... a)
{
    const u32 pr = ((NW == kWordsA ? (g_rc_prio & 1u) : NW == kWordsB ? (g_rc_prio & 4u) : 0u) ? 8u : 0u) | ((NW == kWordsA ? (g_rc_prio & 8u) : NW == kWordsB ? (g_rc_prio & 16u) : 0u) ? 16u : 0u);
    // (k_rcwalk3's walking wave alone at raised priority where k_rcwalk2w raises every wave: 39.2 against 39.5 ms per step, tools/r05/run5.sh)
    if (g_rc_ww == 1 && NW == kWordsA && nitems >= g_rc3_min) {
        if (g_rc3_band == 16) hipLaunchKernelGGL((k_rcwalk3<NW, TW, COLS, MAXOPS, 16>), dim3((nitems + 63) / 64), dim3(128), 0, s, a..., (pr & 8u) ? 16u : pr);
        else hipLaunchKernelGGL((k_rcwalk3<NW, TW, COLS, MAXOPS, 32>), dim3((nitems + 63) / 64), dim3(128), 0, s, a..., (pr & 8u) ? 16u : pr);
    }
    else if (g_rc_ww >= 2 && g_rc3_band == 16) hipLaunchKernelGGL((k_rcwalk3<NW, TW, COLS, MAXOPS, 16>), dim3((nitems + 63) / 64), dim3(128), 0, s, a..., pr);
    else if (g_rc_ww >= 2) hipLaunchKernelGGL((k_rcwalk3<NW, TW, COLS, MAXOPS, 32>), dim3((nitems + 63) / 64), dim3(128), 0, s, a..., pr);
    else if (g_rc_ww) hipLaunchKernelGGL((k_rcwalk2w<NW, TW, COLS, MAXOPS>), dim3((nitems + 63) / 64), dim3(256), 0, s, a..., g_rc_prefetch | g_rc_dbg | pr);
    else hipLaunchKernelGGL((k_rcwalk2<NW, TW, COLS, MAXOPS>), dim3((nitems + 15) / 16), dim3(64), 0, s, a...);
}

// Tuning / test knobs of ONE context: read from the environment when it is created (knobs.h), defaults otherwise.
void read_knobs(necat::Knobs& K)
{
    auto num = [](const char* name, unsigned long long dflt) { const char* e = getenv(name); return e ? strtoull(e, nullptr, 10) : dflt; };
    K.coop_threshold = (u32)num("NECAT_COOP_THRESHOLD", 0xffffffffu);
    K.seed_budget = num("NECAT_SEED_BUDGET", 48ULL << 20);
    K.single_pass = (u32)num("NECAT_SINGLE_PASS", 4096);
    K.tail_fused = (u32)num("NECAT_TAIL_FUSED", 512);
    K.asm_lane = (int)num("NECAT_ASM_LANE", 0);
    K.walk_wave = (u32)num("NECAT_WALK_WAVE", 12288);
    K.rcwalk = (u32)num("NECAT_RCWALK", 512);
    K.rc_carry = (u32)num("NECAT_RC_CARRY", 1);
    K.asm_rc = (u32)num("NECAT_ASM_RC", 1);
    K.rc_ww = (u32)num("NECAT_RC_WW", 1);
    K.rc3_min = (u32)num("NECAT_RC3_MIN", 160000);
    K.rc3_band = num("NECAT_RC3_BAND", 32) == 16 ? 16u : 32u;
    K.rc_prefetch = (u32)num("NECAT_RC_PREFETCH", 0);
    K.rc_dbg = (u32)num("NECAT_RC_DBG", 0) & 6u;
    K.rc_fastb = (u32)num("NECAT_RC_FASTB", 1);
    K.ck_post = (u32)num("NECAT_CK_POST", 1);
    K.rc_prio = (u32)num("NECAT_RC_PRIO", 1);
    K.rc_pipe = (u32)std::min<unsigned long long>(8, std::max<unsigned long long>(1, num("NECAT_RC_PIPE", 1))); K.rc_pipe_min = (u32)num("NECAT_RC_PIPE_MIN", 49152);
    K.rc_merge = (u32)num("NECAT_RC_MERGE", 1);
    K.ck_lds = (u32)num("NECAT_CK_LDS", 0);
    K.rc_listb = K.rc_carry ? (u32)num("NECAT_RC_LISTB", 1) : 0u;
    K.rc_ragged = K.rc_carry ? (u32)num("NECAT_RC_RAGGED", 1) : 0u;
    K.rc_pool = (size_t)std::max<unsigned long long>(1, num("NECAT_RC_POOL_MB", 8192)) << 20;
    K.rc_maxdist = (int)num("NECAT_RC_MAXDIST", K.rc_carry ? 1 << 20 : kRcMaxDist);
    if (!K.rc_carry) K.rc_maxdist = std::min(K.rc_maxdist, kRcMaxDist);
    K.batch_cap = (u32)std::max<unsigned long long>(64, num("NECAT_BATCH", 786432));
    K.ext_overlap = (u32)num("NECAT_EXT_OVERLAP", 1);
    K.ext_overlap_min = (u32)num("NECAT_EXT_OVERLAP_MIN", 0);
    K.ext_overlap_pct = (u32)std::min<unsigned long long>(100, num("NECAT_EXT_OVERLAP_PCT", 100));
    K.ext_overlap_order = (u32)num("NECAT_EXT_ORDER", 1);
    K.ext_overlap_split = (u32)std::min<unsigned long long>(95, std::max<unsigned long long>(5, num("NECAT_EXT_OVERLAP_SPLIT", 20)));
    K.index_lds = (int)num("NECAT_INDEX_LDS", 1);
    K.split_threads = num("NECAT_SPLIT_THREADS", 512) == 256 ? 256 : 512;
    K.seed_wave = (int)num("NECAT_SEED_WAVE", 1);
    K.seed_kst = (int)num("NECAT_SEED_KST", 1);
    K.trace = (int)num("NECAT_TRACE", 0);
    K.coop_filter = (int)num("NECAT_COOP_FILTER", 1);
    K.sort_b = (int)num("NECAT_SORT_B", 1);
    K.dbg = (int)num("NECAT_DBG", 0);
    K.fast = (int)num("NECAT_FAST", 1);
    K.band_pool = (size_t)num("NECAT_BAND_POOL_MB", 16384) << 20;   // 16 GB = 250 k list-A blocks per launch: as efficient as the whole list, and the first call does not allocate 50 - 100 GB
    K.fast16 = (int)num("NECAT_FAST16", 0);      // measured: no gain on the bench workload, off by default
    K.walk = (int)num("NECAT_WALK", 0);      // 0: reference formulation (default until the restated walk wins), 1: walk_block, 2: walk_block without record prefetch
    K.cns_spec_extra = getenv("NECAT_CNS_SPEC_EXTRA") ? atoi(getenv("NECAT_CNS_SPEC_EXTRA")) : 1;
    K.cns_spec_cover = (int)num("NECAT_CNS_SPEC", 12);     // 0 = adaptive
}

double wall_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

double ev_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0; return ms; }

}  // namespace

extern "C" {

void necat_default_options(necat_map_options* o)
{   // map_options.c:12-28 (sDefaultPairwiseMapingOptions)
    o->kmer_size = 15; o->scan_window = 10; o->kmer_cnt_cutoff = 500; o->block_size = 2000;
    o->block_score_cutoff = 3; o->num_candidates = 500; o->align_size_cutoff = 500;
    o->ddfs_cutoff = 0.25; o->error = 0.5; o->num_output = 500; o->num_threads = 1;
    o->job = 1; o->binary_output = 0; o->use_hdr_as_id = 1;
}

int necat_ctx_create(int device_id, necat_ctx** out)
{
    if (!out) return NECAT_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NECAT_ERR_DEVICE;
    if (device_id < 0 || device_id >= ndev) return NECAT_ERR_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return NECAT_ERR_DEVICE;
    necat_ctx* ctx = new necat_ctx();
    ctx->device = device_id;
    read_knobs(ctx->knobs);
    memset(&ctx->tm, 0, sizeof ctx->tm);
    memset(&ctx->shard_tm, 0, sizeof ctx->shard_tm);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        snprintf(ctx->devname, sizeof ctx->devname, "%s (%s), %d CUs", prop.name, prop.gcnArchName, prop.multiProcessorCount);
        ctx->num_cu = prop.multiProcessorCount;
    }
    // one stream now; the extension's streams when the first extension runs (a stream costs 8 - 15 ms to create: a candidates-only
    // process never pays for them)
    if (hipStreamCreate(&ctx->stream) != hipSuccess) { delete ctx; return NECAT_ERR_DEVICE; }
    for (int i = 0; i < kNumEvents; ++i) if (hipEventCreate(&ctx->ev[i]) != hipSuccess) { delete ctx; return NECAT_ERR_DEVICE; }
    // the list sizes of the extension rounds reach the host through this pinned ring (RoundPub, ext_kernels.h)
    if (hipHostMalloc(&ctx->round_ring, 2 * kRoundRing * sizeof(RoundPub), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&ctx->round_ring_dev, ctx->round_ring, 0) != hipSuccess) { delete ctx; return NECAT_ERR_DEVICE; }
    memset(ctx->round_ring, 0, 2 * kRoundRing * sizeof(RoundPub));      // (lane 1's half: ExtLane1)
    *out = ctx;
    return NECAT_OK;
}

void necat_ctx_destroy(necat_ctx* ctx)
{
    KnobScope knob_scope_(ctx);
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (g_trace & 2) {        // what the context holds at its end: the arenas of 64 MB and more (ScratchId : MB)
        size_t sum = 0; char own[1024]; int at = 0; own[0] = 0;
        for (int i = 0; i < (int)(sizeof ctx->scratch / sizeof ctx->scratch[0]); ++i) {
            sum += ctx->scratch[i].cap;
            if (ctx->scratch[i].cap >= ((size_t)64 << 20) && at < 980) at += snprintf(own + at, sizeof own - (size_t)at, " %d:%zu", i, ctx->scratch[i].cap >> 20);
        }
        fprintf(stderr, "[necat] context arenas at destroy: %zu MB in all; [id:MB]%s\n", sum >> 20, own);
    }
    if (ctx->stream_copy) (void)hipStreamSynchronize(ctx->stream_copy);
    for (auto& b : ctx->scratch) if (b.p) (void)hipFree(b.p);
    for (auto& b : ctx->idx_cache) if (b.p) (void)hipFree(b.p);
    for (auto& b : ctx->lane1.buf) if (b.p) (void)hipFree(b.p);
    if (ctx->lane1.ready) for (int i = 0; i < kNumEvents; ++i) (void)hipEventDestroy(ctx->lane1.ev[i]);
    for (hipStream_t st : ctx->lane1.st) if (st) (void)hipStreamDestroy(st);
    delete (cns::Scratch*)ctx->cns_scratch;
    if (ctx->pin_plan) (void)hipHostFree(ctx->pin_plan);
    for (int i = 0; i < kNumEvents; ++i) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->round_ring) (void)hipHostFree(ctx->round_ring);
    if (ctx->serial_streams) ctx->stream_a = ctx->stream_b = ctx->stream_c = ctx->stream_d = nullptr;      // aliases of ctx->stream (NECAT_SERIAL)
    for (hipStream_t st : {ctx->stream, ctx->stream_a, ctx->stream_b, ctx->stream_c, ctx->stream_d, ctx->stream_copy}) if (st) (void)hipStreamDestroy(st);
    delete ctx;
}

void necat_ctx_trim(necat_ctx* ctx)
{
    KnobScope knob_scope_(ctx);
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& b : ctx->scratch) if (b.p) { (void)hipFree(b.p); b = DevBuf(); }
    for (auto& b : ctx->idx_cache) if (b.p) { (void)hipFree(b.p); b = DevBuf(); }
    for (auto& b : ctx->lane1.buf) if (b.p) { (void)hipFree(b.p); b = DevBuf(); }
    ctx->seed_ht_ptr = nullptr; ctx->seed_ht_clean = 0; ctx->seed_ht_cap = 0;
}

const char* necat_last_error(const necat_ctx* ctx) { return ctx ? ctx->err : "no context"; }

int necat_device_name(const necat_ctx* ctx, char* buf, size_t n)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !buf || !n) return NECAT_ERR_ARG;
    snprintf(buf, n, "%s", ctx->devname);
    return NECAT_OK;
}

int necat_get_timings(const necat_ctx* ctx, necat_timings* t)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !t) return NECAT_ERR_ARG;
    *t = ctx->tm;
    return NECAT_OK;
}

// Result blocks handed to the caller.  Big ones are pinned host memory (the device copies straight into
// them) and are recycled: necat_free() parks up to kPoolBlocks of them for the next call instead of
// returning 20 MB of freshly faulted pages to the OS every pass.
namespace {
struct ResultPool {
    std::mutex mu;
    std::unordered_map<void*, size_t> live;              // pinned blocks currently owned by callers
    std::vector<std::pair<void*, size_t>> parked;         // never released at exit: the HIP runtime may be gone by then
};
ResultPool g_results;
constexpr size_t kPinnedMin = 256 << 10, kPoolBlocks = 16, kPoolBytes = (size_t)8 << 30;   // parked: <= 16 blocks, <= 8 GiB

void* result_alloc(size_t bytes)
{
    if (bytes < kPinnedMin) return malloc(bytes ? bytes : 1);
    std::lock_guard<std::mutex> lk(g_results.mu);
    int best = -1;
    for (int i = 0; i < (int)g_results.parked.size(); ++i) {
        const size_t sz = g_results.parked[i].second;
        if (sz >= bytes && sz <= 2 * bytes + (1 << 20) && (best < 0 || sz < g_results.parked[best].second)) best = i;
    }
    void* p = nullptr; size_t sz = 0;
    if (best >= 0) { p = g_results.parked[best].first; sz = g_results.parked[best].second; g_results.parked.erase(g_results.parked.begin() + best); }
    else {
        sz = (bytes + (bytes >> 3) + 4095) & ~(size_t)4095;          // a little slack so the next pass fits too
        if (hipHostMalloc(&p, sz, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return malloc(bytes); }
    }
    g_results.live[p] = sz;
    return p;
}
}  // namespace

void necat_free(void* p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_results.mu);
        auto it = g_results.live.find(p);
        if (it != g_results.live.end()) {
            const size_t sz = it->second;
            g_results.live.erase(it);
            // park it; when the pool is full the smallest blocks go first (pinning costs ~0.15 s per GB, so the big
            // ones are the ones worth keeping)
            g_results.parked.emplace_back(p, sz);
            size_t held = 0;
            for (auto& b : g_results.parked) held += b.second;
            while (g_results.parked.size() > kPoolBlocks || (held > kPoolBytes && !g_results.parked.empty())) {
                size_t k = 0;
                for (size_t i = 1; i < g_results.parked.size(); ++i) if (g_results.parked[i].second < g_results.parked[k].second) k = i;
                held -= g_results.parked[k].second;
                (void)hipHostFree(g_results.parked[k].first);
                g_results.parked.erase(g_results.parked.begin() + k);
            }
            return;
        }
    }
    free(p);
}

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
    const int rc = index_build_body(ctx, comm, ref, kmer_size, max_occ, ix);
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

// ------------------------------------------------------------------------------------------ seeding

namespace {
// candidates left on the device for necat_map_pair: array in ascending read order + the first candidate of
// every read that has any (the groups of the containment filter) + the total
struct DevCands { const necat_candidate* d = nullptr; uint64_t n = 0; std::vector<u64> group_off; };

void fill_groups(DevCands* dev, const std::vector<u64>& by_read, u32 nreads)
{
    dev->group_off.clear();
    for (u32 r = 0; r < nreads; ++r) if (by_read[r + 1] > by_read[r]) dev->group_off.push_back(by_read[r]);
    dev->group_off.push_back(by_read[nreads]);
    if (dev->group_off.size() == 1) dev->group_off.insert(dev->group_off.begin(), 0);
}

// the query reads one rank of a multi-GPU job processes: chunks of `chunk` reads, chunk c in slot c % nparts
// (one rank of a sharded call: slots [rank, rank + 1) of nranks; a share of a scheduled volume pair: slots [lo, hi) of `nparts`
// - interleaved either way, because a read late in a volume sees more subjects in the self pair, word_finder.c:121-127)
struct ReadSel {
    int lo = 0, hi = 1, nparts = 1, chunk = 64;
    bool always = false;            // apply the slot test even when nparts == 1 (an empty share selects nothing)
    bool has(u32 r) const
    {
        if (nparts <= 1 && !always) return true;
        const int sl = (int)((r / (u32)chunk) % (u32)nparts);
        return sl >= lo && sl < hi;
    }
};

int find_impl(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
              int read_start_id, int ref_start_id, int pairwise, const necat_map_options* opt,
              necat_candidate** out, uint64_t* n_out, DevCands* dev, const ReadSel* sel = nullptr)
{
    if (opt->kmer_size != ix->k) return set_err(ctx, NECAT_ERR_ARG, "index was built for k=%d, options say %d", ix->k, opt->kmer_size);
    if (opt->scan_window < 1 || opt->block_size < 1 || opt->block_size > 32767)
        return set_err(ctx, NECAT_ERR_ARG, "scan_window/block_size out of range (block offsets are 16-bit, word_finder_aux.h:21)");
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const u32 nreads = (u32)reads->nseq;
    if (nreads == 0) return NECAT_OK;
    ArenaUse in_use(ctx, {SC_SEED_POOL, SC_SEED_CHAIN, SC_SEED_OUT, SC_SEED_HT, SC_SEED_META});      // (buf_ensure_lend: held for the length of this call)
    DevVolume dref = dev_view(ref), drd = dev_view(reads);
    NECAT_HIP(ctx, hipEventRecord(ctx->ev[0], s));
    int rc;
    auto t_prev = std::chrono::steady_clock::now();
    auto tick = [&](const char* what) {
        if (!(g_trace & 2)) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[necat] seeding %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // ---- pass 1: hit counts per read-strand
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_MISC], (size_t)nreads * 8 + 128))) return rc;
    u32* d_hits = (u32*)ctx->scratch[SC_MISC].p;
    int* d_err = (int*)((char*)ctx->scratch[SC_MISC].p + (((size_t)nreads * 8 + 63) & ~(size_t)63));   // error flag of the seeding kernels
    // the table words k_seed_hits fetches are kept for the collection pass (seed_kst_base): one lookup per sampled k-mer, not two
    u64* d_kst = nullptr;
    if (g_seed_wave && g_seed_kst) {
        if ((rc = buf_ensure(ctx, ctx->scratch[SC_SEED_KST], 2 * (reads->nbases / (u64)opt->scan_window + nreads + 2) * 8))) return rc;
        d_kst = (u64*)ctx->scratch[SC_SEED_KST].p;
    }
    hipLaunchKernelGGL(k_seed_hits, dim3(grid_for((u64)nreads * 64, 256)), dim3(256), 0, s, drd, index_view(ix),
                       opt->kmer_size, opt->scan_window, 0u, nreads, d_hits, d_kst);
    NECAT_CHECK_LAUNCH(ctx, "k_seed_hits");
    // pinned host scratch: [hits: 2 u32 per read][order: u32 per read][SeedMeta per read] - pageable copies cost more than the plan
    {
        const size_t need = (size_t)nreads * (8 + 4 + sizeof(SeedMeta)) + 256;
        if (need > ctx->pin_plan_cap) {
            if (ctx->pin_plan) (void)hipHostFree(ctx->pin_plan);
            ctx->pin_plan = nullptr; ctx->pin_plan_cap = 0;
            if (hipHostMalloc(&ctx->pin_plan, need + need / 4, hipHostMallocDefault) != hipSuccess) return set_err(ctx, NECAT_ERR_MEMORY, "pinned host scratch (%zu bytes)", need);
            ctx->pin_plan_cap = need + need / 4;
        }
    }
    u32* hits = (u32*)ctx->pin_plan;
    u32* order = hits + (size_t)nreads * 2;
    SeedMeta* meta_all = (SeedMeta*)(((uintptr_t)(order + nreads) + 63) & ~(uintptr_t)63);
    NECAT_HIP(ctx, hipMemcpyAsync(hits, d_hits, (size_t)nreads * 8, hipMemcpyDeviceToHost, s));
    NECAT_HIP(ctx, hipStreamSynchronize(s));
    tick("hits kernel + copy");
    // ---- plan: reads in descending work order, chunks bounded by a scratch budget
    u32 nsel = 0;
    {
        // descending work, ascending read id inside equal work (only this rank's reads): a stable LSD radix sort of the
        // complemented hit counts, 3 x 11 bits (std::sort of the same keys took ~1 ms for 23 k reads)
        std::vector<u32> key(nreads), ida(nreads), idb(nreads);
        for (u32 r = 0; r < nreads; ++r)
            if (!sel || sel->has(r)) { key[r] = 0xffffffffu - std::max(hits[2 * (size_t)r], hits[2 * (size_t)r + 1]); ida[nsel++] = r; }
        u32* src = ida.data(); u32* dst = idb.data();
        for (int pass = 0; pass < 3; ++pass) {
            const int sh = 11 * pass;
            u32 cnt[2049] = {0};
            for (u32 i = 0; i < nsel; ++i) ++cnt[((key[src[i]] >> sh) & 2047u) + 1];
            for (int b = 0; b < 2048; ++b) cnt[b + 1] += cnt[b];
            for (u32 i = 0; i < nsel; ++i) dst[cnt[(key[src[i]] >> sh) & 2047u]++] = src[i];
            std::swap(src, dst);
        }
        for (u32 i = 0; i < nsel; ++i) order[i] = src[i];
    }
    ctx->shard_tm.reads_local = nsel;
    {   // the terms of SURVEY 8d's B_seed for this call (bench.py: roofline_seed)
        u64 lk = 0, ht = 0, bs = 0;
        const bool have_off = reads->h_seq_off.size() == (size_t)nreads + 1;
        for (u32 i = 0; i < nsel; ++i) {
            const u32 r = order[i];
            ht += (u64)hits[2 * (size_t)r] + hits[2 * (size_t)r + 1];
            if (have_off) { const u64 L = reads->h_seq_off[r + 1] - reads->h_seq_off[r]; bs += L; if (L >= (u64)opt->kmer_size) lk += (L - (u64)opt->kmer_size) / (u64)opt->scan_window + 1; }
        }
        ctx->tm.seed_bases = 2 * bs; ctx->tm.seed_lookups = 2 * lk; ctx->tm.seed_hits = ht; ctx->tm.seed_cands = 0;
    }
    if (nsel == 0) {
        if (dev) { dev->n = 0; dev->d = nullptr; dev->group_off.assign(2, 0); }
        else { *out = (necat_candidate*)result_alloc(sizeof(necat_candidate)); *n_out = 0; }
        ctx->tm.seed_ms = 0;
        return NECAT_OK;
    }
    const u64 budget_hits = g_seed_budget;   // default ~48 M pool blocks (~13 GB of SBlocks) per chunk
    SeedParams P;
    P.k = opt->kmer_size; P.z = opt->scan_window; P.block_size = opt->block_size; P.s_cutoff = opt->block_score_cutoff;
    P.align_cutoff = opt->align_size_cutoff; P.num_candidates = opt->num_candidates; P.job = opt->job; P.pairwise = pairwise;
    P.read_start_id = read_start_id; P.ref_start_id = ref_start_id;
    P.debug_phase = getenv("NECAT_SEED_DEBUG") ? atoi(getenv("NECAT_SEED_DEBUG")) : 0;
    P.chain_wave = getenv("NECAT_CHAIN_WAVE") ? atoi(getenv("NECAT_CHAIN_WAVE")) : 1;
    NECAT_HIP(ctx, hipMemsetAsync(d_err, 0, 4, s));
    u32 pos = 0;
    std::vector<i32> ncands_by_order(nsel, 0);
    // every chunk's compacted candidates stay on the device (SC_SEED_ALL), in ORDER-index space
    u64 packed_total = 0;
    std::vector<u64> packed_off(nsel + 1, 0);
    while (pos < nsel) {
        u64 acc = 0; u32 hi = pos;
        auto both = [&](u32 r) { return (u64)hits[2 * (size_t)r] + hits[2 * (size_t)r + 1] + 2; };
        while (hi < nsel && (hi == pos || acc + both(order[hi]) <= budget_hits)) { acc += both(order[hi]); ++hi; }
        const u32 n = hi - pos;
        SeedMeta* meta = meta_all + pos;
        u64 ht_tot = 0, pool_tot = 0, chain_tot = 0, out_tot = 0;
        for (u32 i = 0; i < n; ++i) {
            const u32 r = order[pos + i];
            SeedMeta& m = meta[i];
            for (int st = 0; st < 2; ++st) {
                const u64 H = std::max<u64>(1, hits[2 * (size_t)r + st]);
                u64 cap = 4; while (cap < 2 * H) cap <<= 1;
                m.ht_off[st] = ht_tot; m.ht_mask[st] = (u32)(cap - 1); ht_tot += cap;
                m.pool_off[st] = pool_tot; m.pool_cap[st] = (u32)H; pool_tot += H;
                // chain scratch per strand: the two strands of a read are evaluated by two waves at the same time
                m.chain_off[st] = chain_tot; m.cs_cap[st] = (u32)(H + 1); chain_tot += H + 1;
            }
            const u64 oc = (u64)hits[2 * (size_t)r] + hits[2 * (size_t)r + 1] + 2;
            m.out_off = out_tot; m.out_cap = (u32)oc; m.out_cap0 = hits[2 * (size_t)r] + 1; out_tot += oc;
        }
        if ((rc = buf_ensure(ctx, ctx->scratch[SC_SEED_META], n * sizeof(SeedMeta) + (size_t)n * (8 + 4 + 4 + 8 + 8) + 64)) ||
            (rc = buf_ensure(ctx, ctx->scratch[SC_SEED_HT], ht_tot * 8)) ||
            // (the block pool and the chain scratch take over the index build's split buffers, idle until the next build: runtime.h)
            (rc = buf_ensure_lend(ctx, SC_SEED_POOL, pool_tot * sizeof(SBlock), {SC_PART2, SC_TMPLIST})) ||
            (rc = buf_ensure_lend(ctx, SC_SEED_CHAIN, chain_tot * (8 + 16 + 8 + sizeof(DevCand)), {SC_TMPLIST, SC_PART2})) ||
            (rc = buf_ensure_lend(ctx, SC_SEED_OUT, out_tot * sizeof(DevCand), {SC_TMPLIST, SC_PART2}))) { return rc; }
        tick("plan + buffers");
        char* mb = (char*)ctx->scratch[SC_SEED_META].p;
        SeedMeta* d_meta = (SeedMeta*)mb; mb += n * sizeof(SeedMeta);
        u64* d_final = (u64*)mb; mb += (size_t)n * 8;
        i32* d_nblk = (i32*)mb; mb += (size_t)n * 8;
        u32* d_order = (u32*)mb; mb += (size_t)n * 4;
        i32* d_ncand = (i32*)mb; mb += (size_t)n * 4;
        i32* d_nstrand = (i32*)mb;
        SeedArenas A;
        A.ht = (u64*)ctx->scratch[SC_SEED_HT].p;
        A.pool = (SBlock*)ctx->scratch[SC_SEED_POOL].p;
        char* cb = (char*)ctx->scratch[SC_SEED_CHAIN].p;
        A.cs = (u64*)cb; cb += chain_tot * 8;
        A.u = (u64*)cb; cb += chain_tot * 8;
        A.lcan = (DevCand*)cb; cb += chain_tot * sizeof(DevCand);
        A.f = (i32*)cb; cb += chain_tot * 4; A.p = (i32*)cb; cb += chain_tot * 4; A.t = (i32*)cb; cb += chain_tot * 4; A.v = (i32*)cb;
        A.out = (DevCand*)ctx->scratch[SC_SEED_OUT].p;
        NECAT_HIP(ctx, hipMemcpyAsync(d_meta, meta, n * sizeof(SeedMeta), hipMemcpyHostToDevice, s));
        NECAT_HIP(ctx, hipMemcpyAsync(d_order, order + pos, (size_t)n * 4, hipMemcpyHostToDevice, s));
        // the hash arena is all-empty between calls (k_seed_clear below): filled only when it is new or a failed call left it dirty
        // (only the stretch this chunk uses beyond what is known clean: a fresh 13 GB arena is not filled for a 0.3 GB chunk)
        if (ctx->seed_ht_ptr != ctx->scratch[SC_SEED_HT].p || ctx->seed_ht_cap != ctx->scratch[SC_SEED_HT].cap) {      // a new allocation (also one at the old address)
            ctx->seed_ht_ptr = ctx->scratch[SC_SEED_HT].p; ctx->seed_ht_cap = ctx->scratch[SC_SEED_HT].cap; ctx->seed_ht_clean = 0;
        }
        const size_t ht_clean_before = ctx->seed_ht_clean;
        if (ht_clean_before < ht_tot * 8) NECAT_HIP(ctx, hipMemsetAsync((char*)A.ht + ht_clean_before, 0xFF, ht_tot * 8 - ht_clean_before, s));
        const size_t ht_clean_after = std::max<size_t>(ht_clean_before, ht_tot * 8);
        ctx->seed_ht_clean = 0;      // in use: clean again once this chunk's kernels (k_seed_clear last) are known to have run
        if (g_seed_wave)
            hipLaunchKernelGGL(k_seed_collect_wave, dim3(2 * n), dim3(64), 0, s, dref, drd, index_view(ix), (const u64*)ix->offset_list,
                               P, (const u32*)d_order, (const SeedMeta*)d_meta, n, A, d_nblk, d_err, (const u64*)d_kst);
        else
            hipLaunchKernelGGL(k_seed_collect, dim3(grid_for((u64)2 * n, 64)), dim3(64), 0, s, dref, drd, index_view(ix), (const u64*)ix->offset_list,
                               P, (const u32*)d_order, (const SeedMeta*)d_meta, n, A, d_nblk, d_err);
        NECAT_CHECK_LAUNCH(ctx, "k_seed_collect");
        static const bool fused_clear = !getenv("NECAT_SEED_CLEAR_KERNEL");        // (A/B: the slots cleared by a launch of their own, as in round 3)
        hipLaunchKernelGGL(k_seed_eval, dim3(2 * n), dim3(64), 0, s, dref, drd, P, (const u32*)d_order, (const SeedMeta*)d_meta, n, A,
                           (const i32*)d_nblk, d_nstrand, d_err, fused_clear && P.debug_phase != 1 ? 1 : 0);
        NECAT_CHECK_LAUNCH(ctx, "k_seed_eval");
        if (!fused_clear || P.debug_phase == 1) {
            hipLaunchKernelGGL(k_seed_clear, dim3(2 * n), dim3(64), 0, s, (const SeedMeta*)d_meta, n, A, (const i32*)d_nblk);
            NECAT_CHECK_LAUNCH(ctx, "k_seed_clear");
        }
        hipLaunchKernelGGL(k_seed_finish, dim3(grid_for(n, 64)), dim3(64), 0, s, P, (const SeedMeta*)d_meta, n, A, (const i32*)d_nstrand, d_ncand);
        NECAT_CHECK_LAUNCH(ctx, "k_seed_finish");
        std::vector<i32> nc(n);
        NECAT_HIP(ctx, hipMemcpyAsync(nc.data(), d_ncand, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        int herr = 0;
        NECAT_HIP(ctx, hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, s));
        NECAT_HIP(ctx, hipStreamSynchronize(s));
        if (herr) { return set_err(ctx, NECAT_ERR_CAPACITY, "seeding scratch overflow (code %d)", herr); }
        ctx->seed_ht_clean = ht_clean_after;
#ifdef NECAT_SEED_PROF
        {   // tools/seed_prof.sh: cycles of lane 0 per phase of k_seed_eval, summed over the waves
            unsigned long long h[32], z[32] = {0};
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_seed_prof), sizeof h) == hipSuccess) {
                static const char* nm[10] = {"block test", "A seed lists", "B vote", "C anchor", "D gather", "E sort", "emit (lane 0)", "chain DP (wave)", "evaluations", "exit"};
                unsigned long long tot = 0; for (int q = 0; q < 10; ++q) if (q != 8) tot += h[q];
                for (int q = 0; q < 10; ++q) fprintf(stderr, "[seed prof] %-16s %14llu %5.1f %%\n", nm[q], h[q], q == 8 ? 0.0 : 100.0 * h[q] / (double)tot);
                fprintf(stderr, "[seed prof] longest wave %llu cycles, most evaluations in a wave %llu\n", h[10], h[11]);
                if (h[26]) for (int q = 0; q < 10; ++q) fprintf(stderr, "[seed prof] waves over 3 M cycles (%llu): %-16s %12llu per wave\n", h[26], nm[q], h[16 + q] / h[26]);
            }
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_seed_prof), z, sizeof z);
        }
#endif
        tick("collect + eval kernels");
        if (pos == 0 && hi == nsel) {
            // the usual case, one chunk: pack on the device straight into ascending read order and copy
            // into the (pinned) result block
            std::vector<u64> by_read((size_t)nreads + 1, 0), foff(n);
            for (u32 i = 0; i < n; ++i) by_read[order[i] + 1] = (u64)nc[i];
            for (u32 r = 0; r < nreads; ++r) by_read[r + 1] += by_read[r];
            for (u32 i = 0; i < n; ++i) foff[i] = by_read[order[i]];
            const u64 tot = by_read[nreads];
            necat_candidate* res = dev ? nullptr : (necat_candidate*)result_alloc(std::max<u64>(1, tot) * sizeof(necat_candidate));
            if (!dev && !res) { return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"); }
            if (dev) { dev->n = tot; fill_groups(dev, by_read, nreads); }
            if (tot) {
                if ((rc = buf_ensure(ctx, ctx->scratch[SC_SEED_FINAL], tot * sizeof(necat_candidate)))) { necat_free(res); return rc; }
                necat_candidate* d_dst = (necat_candidate*)ctx->scratch[SC_SEED_FINAL].p;
                hipError_t e1 = hipMemcpyAsync(d_final, foff.data(), (size_t)n * 8, hipMemcpyHostToDevice, s);
                hipLaunchKernelGGL(k_pack_cands, dim3(grid_for((u64)n * 64, 256)), dim3(256), 0, s, (const DevCand*)A.out, (const SeedMeta*)d_meta,
                                   (const i32*)d_ncand, (const u64*)d_final, n, read_start_id, ref_start_id, d_dst);
                hipError_t e2 = hipGetLastError();
                if (dev) dev->d = d_dst;
                hipError_t e3 = dev ? hipSuccess : hipMemcpyAsync(res, d_dst, tot * sizeof(necat_candidate), hipMemcpyDeviceToHost, s);
                hipError_t e4 = hipEventRecord(ctx->ev[1], s);
                hipError_t e5 = hipStreamSynchronize(s);
                for (hipError_t e : {e1, e2, e3, e4, e5})
                    if (e != hipSuccess) { necat_free(res); return set_err(ctx, NECAT_ERR_DEVICE, "seeding result copy: %s", hipGetErrorString(e)); }
            } else { NECAT_HIP(ctx, hipEventRecord(ctx->ev[1], s)); NECAT_HIP(ctx, hipStreamSynchronize(s)); }
            ctx->tm.seed_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
            ctx->tm.seed_cands = tot;
            tick("pack + copy to host");
            if (!dev) { *out = res; *n_out = tot; }
            return NECAT_OK;
        }
        // several chunks: pack this chunk's candidates behind the earlier ones, on the device
        std::vector<u64> foff(n + 1, 0);
        for (u32 i = 0; i < n; ++i) foff[i + 1] = foff[i] + (u64)nc[i];
        const u64 tot = foff[n];
        if (tot) {
            if ((rc = buf_grow(ctx, ctx->scratch[SC_SEED_ALL], (packed_total + tot) * sizeof(necat_candidate), packed_total * sizeof(necat_candidate), s))) { return rc; }
            necat_candidate* d_dst = (necat_candidate*)ctx->scratch[SC_SEED_ALL].p + packed_total;
            NECAT_HIP(ctx, hipMemcpyAsync(d_final, foff.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_pack_cands, dim3(grid_for((u64)n * 64, 256)), dim3(256), 0, s, (const DevCand*)A.out, (const SeedMeta*)d_meta,
                               (const i32*)d_ncand, (const u64*)d_final, n, read_start_id, ref_start_id, d_dst);
            NECAT_CHECK_LAUNCH(ctx, "k_pack_cands");
            NECAT_HIP(ctx, hipStreamSynchronize(s));       // foff / nc are host vectors of this iteration
        }
        for (u32 i = 0; i < n; ++i) { ncands_by_order[pos + i] = nc[i]; packed_off[pos + i] = packed_total + foff[i]; }
        packed_total += tot;
        tick("pack");
        pos = hi;
    }
    // ---- ascending read id: one move on the device, one copy into the (pinned) result block
    const u64 total = packed_total;
    necat_candidate* res = dev ? nullptr : (necat_candidate*)result_alloc(std::max<u64>(1, total) * sizeof(necat_candidate));
    if (!dev && !res) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
    std::vector<u64> by_read((size_t)nreads + 1, 0), dst_off(nsel);
    for (u32 i = 0; i < nsel; ++i) by_read[order[i] + 1] = (u64)ncands_by_order[i];
    for (u32 r = 0; r < nreads; ++r) by_read[r + 1] += by_read[r];
    for (u32 i = 0; i < nsel; ++i) dst_off[i] = by_read[order[i]];
    if (dev) { dev->n = total; fill_groups(dev, by_read, nreads); }
    if (total) {
        int rc2;
        if ((rc2 = buf_ensure(ctx, ctx->scratch[SC_SEED_FINAL], total * sizeof(necat_candidate))) ||
            (rc2 = buf_ensure(ctx, ctx->scratch[SC_SEED_META], (size_t)nreads * 20 + 64))) { necat_free(res); return rc2; }
        char* mb = (char*)ctx->scratch[SC_SEED_META].p;
        u64* d_src = (u64*)mb; mb += (size_t)nreads * 8;
        u64* d_dsto = (u64*)mb; mb += (size_t)nreads * 8;
        i32* d_cnt = (i32*)mb;
        necat_candidate* d_fin = (necat_candidate*)ctx->scratch[SC_SEED_FINAL].p;
        hipError_t e[7];
        e[0] = hipMemcpyAsync(d_src, packed_off.data(), (size_t)nsel * 8, hipMemcpyHostToDevice, s);
        e[1] = hipMemcpyAsync(d_dsto, dst_off.data(), (size_t)nsel * 8, hipMemcpyHostToDevice, s);
        e[2] = hipMemcpyAsync(d_cnt, ncands_by_order.data(), (size_t)nsel * 4, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(k_move_cands, dim3(grid_for((u64)nsel * 64, 256)), dim3(256), 0, s, (const necat_candidate*)ctx->scratch[SC_SEED_ALL].p,
                           (const u64*)d_src, (const u64*)d_dsto, (const i32*)d_cnt, nsel, d_fin);
        e[3] = hipGetLastError();
        if (dev) dev->d = d_fin;
        e[4] = dev ? hipSuccess : hipMemcpyAsync(res, d_fin, total * sizeof(necat_candidate), hipMemcpyDeviceToHost, s);
        e[5] = hipEventRecord(ctx->ev[1], s);
        e[6] = hipStreamSynchronize(s);
        for (hipError_t x : e) if (x != hipSuccess) { necat_free(res); return set_err(ctx, NECAT_ERR_DEVICE, "seeding result assembly: %s", hipGetErrorString(x)); }
    } else { NECAT_HIP(ctx, hipEventRecord(ctx->ev[1], s)); NECAT_HIP(ctx, hipStreamSynchronize(s)); }
    ctx->tm.seed_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
    ctx->tm.seed_cands = total;
    tick("assemble in read order");
    if (!dev) { *out = res; *n_out = total; }
    return NECAT_OK;
}
}  // namespace

int necat_find_candidates(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                          int read_start_id, int ref_start_id, int pairwise, const necat_map_options* opt,
                          necat_candidate** out, uint64_t* n_out)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ix || !ref || !reads || !opt || !out || !n_out) return NECAT_ERR_ARG;
    *out = nullptr; *n_out = 0;
    return find_impl(ctx, ix, ref, reads, read_start_id, ref_start_id, pairwise, opt, out, n_out, nullptr);
}

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
    ExtLane1& Q = ctx->lane1;
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
        for (int i = 0; i < kNumEvents; ++i) if (hipEventCreate(&Q.ev[i]) != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "hipEventCreate failed (second extension lane)");
        Q.ready = true;
    }
    DevBuf* S = Q.buf;
    L.tasks = S + LB_TASKS; L.lists = S + LB_LISTS; L.frag = S + LB_FRAG; L.ops = S + LB_OPS; L.res = S + LB_RES; L.mat = S + LB_MAT;
    L.ckpt = S + LB_CKPT; L.wout = S + LB_WOUT; L.ckptb[0] = S + LB_CKPTB; L.ckptb[1] = S + LB_CKPTB2; L.woutb[0] = S + LB_WOUTB; L.woutb[1] = S + LB_WOUTB2;
    L.matb[0] = S + LB_MATB; L.matb[1] = S + LB_MATB2;
    L.sa = Q.st[0]; L.sb[0] = Q.st[1]; L.sb[1] = Q.st[2]; L.sd = Q.st[3];
    L.ev = Q.ev;
    L.ring = (volatile RoundPub*)ctx->round_ring + kRoundRing; L.ring_dev = (RoundPub*)ctx->round_ring_dev + kRoundRing; L.round_seq = &Q.round_seq;
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
            hipLaunchKernelGGL((k_ext_frag<kWordsB, kTWordsB>), dim3(grid_for((u64)gB * 64 * (kWordsB + kTWordsB), 256)), dim3(256), 0, sb,
                               drd, dref, itB, nB, d_nB, 0u, c.fragB[slot], ctl);
            NECAT_CHECK_LAUNCH(ctx, "k_ext_frag<B>");
            NECAT_HIP(ctx, hipEventRecord(c.b0[slot], sb));
            for (u32 lo = 0; lo < nB; lo += rc_chunk) {
                const u32 hi = std::min<u64>((u64)lo + rc_chunk, (u64)gB * 64), cn = std::min(hi, nB) - lo;
                if (g_rc_fastb)
                hipLaunchKernelGGL((k_myers_ckf<kWordsB, kTWordsB, kColsB, 16>), dim3((cn + 3) / 4), dim3(64), 0, sb, itB, nB, d_nB, 0u, (const u64*)c.fragB[slot], ck, hcar, X.error,
                                   c.resB[slot], X.stats, epoch, lo, hi);
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
        hipLaunchKernelGGL((k_ext_frag<kWordsB, kTWordsB>), dim3(grid_for((u64)gB * 64 * (kWordsB + kTWordsB), 256)), dim3(256), 0, sb,
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
        const BlockItem* itA = c.itemsA[cur];
        const u32* d_nA = c.count + 4 * cur;            // [0] full blocks (front of itemsA), [2] the others (back)
        if (r >= 2) NECAT_HIP(ctx, hipStreamWaitEvent(c.sa, c.b2[r & 1], 0));        // B(r - 2) appended to lists[r]
        const u32 epoch = ++ctx->epoch & 0x3fffffu;
        RoundCtl ctl; ctl.count = d_nA; ctl.zero = c.count + 4 * nxt2; ctl.seq = seq0 + r + 1; ctl.pub = ring_dev + (seq0 + r) % kRoundRing;
        hipLaunchKernelGGL((k_ext_frag<kWordsA, kTWordsA>), dim3(grid_for((u64)std::max(gA, 1u) * 64 * (kWordsA + kTWordsA), 256)), dim3(256), 0, c.sa,
                           drd, dref, itA, bound, d_nA, c.cap, c.fragA, ctl);
        NECAT_CHECK_LAUNCH(ctx, "k_ext_frag<A>");
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
            if (!g_rc_ragged) {
                NECAT_HIP(ctx, hipStreamWaitEvent(sd, c.a0[cur], 0));            // the fragments are there
                hipLaunchKernelGGL((k_myers_coop<kWordsA, kTWordsA, kColsA, 8>), dim3(gA * 8), dim3(64), 0, sd, itA, bound, d_nA, c.cap,
                                   (const u64*)c.fragA, slabsA, kSlabA, X.error, c.resA, X.stats, fl_rag, 0u);
                hipLaunchKernelGGL((k_traceback<kWordsA, kTWordsA, kColsA, kOpsA, false, 0>), dim3(gA), dim3(64), 0, sd, itA, bound, d_nA, c.cap,
                                   (const u64*)c.fragA, (const char*)slabsA, kSlabA, (const BlockResult*)c.resA, c.opsA, c.tasks, X.tail_match_len,
                                   (i32*)nullptr, X.d_err, next, fl_rag, 0u);
                NECAT_CHECK_LAUNCH(ctx, "k_myers / k_traceback<A, ragged>");
            }
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
                                       (merged ? fl_all : epoch) | (g_ck_post ? 0u : 1u << 24) | (g_rc_prio & 2u ? 1u << 23 : 0u));
                else
                    hipLaunchKernelGGL((k_myers_ck<kWordsA, kTWordsA, false>), dim3((cn + 7) / 8), dim3(64), 0, c.sa, itA, d_nA, c.cap, (const u64*)c.fragA, ck, hcar, X.error, c.resA, X.stats, g_rc_maxdist, lo, hi, epoch);
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
                    hipLaunchKernelGGL((k_rcwalk4<kWordsA, kTWordsA, kOpsA>), dim3((cn + 15) / 16), dim3(64), 0, c.sa, itA, d_nA, c.cap, (const u64*)c.fragA, (const ulonglong2*)ck,
                                       (const BlockResult*)c.resA, (const ExtTask*)c.tasks, X.task_ops ? 1 : 0, X.tail_match_len, c.opsA, wo, X.stats, X.d_err, lo, hi);
                NECAT_CHECK_LAUNCH(ctx, "k_rcwalk");
            }
            NECAT_HIP(ctx, hipEventRecord(L.ev[26 + (r & 3)], piped ? sd : c.sa));       // a1 -> this: the walk kernel alone (account_a; of the last chunk, normally the only one)
            if (piped) NECAT_HIP(ctx, hipStreamWaitEvent(c.sa, L.ev[26 + (r & 3)], 0));          // the finishing kernel reads what the walks left
            if (wide_possible) {
                NECAT_HIP(ctx, hipStreamWaitEvent(sd, c.a1[cur], 0));            // k_myers_ck has flagged the wide blocks
                hipLaunchKernelGGL((k_myers_coop<kWordsA, kTWordsA, kColsA, 8>), dim3(gA * 8), dim3(64), 0, sd, itA, bound, d_nA, c.cap,
                                   (const u64*)c.fragA, slabsA, kSlabA, X.error, c.resA, X.stats, fl_wide, 0u);
                hipLaunchKernelGGL((k_traceback<kWordsA, kTWordsA, kColsA, kOpsA, false, 0>), dim3(gA), dim3(64), 0, sd, itA, bound, d_nA, c.cap,
                                   (const u64*)c.fragA, (const char*)slabsA, kSlabA, (const BlockResult*)c.resA, c.opsA, c.tasks, X.tail_match_len,
                                   (i32*)nullptr, X.d_err, next, fl_wide, 0u);
                NECAT_CHECK_LAUNCH(ctx, "k_myers / k_traceback<A, wide>");
            }
            if (!g_rc_ragged || wide_possible) NECAT_HIP(ctx, hipEventRecord(L.ev[25], sd));
            if (g_rc_ragged && one_chunk && !(g_rc_merge && g_rc_carry && !getenv("NECAT_RC_CKG_ALL"))) NECAT_HIP(ctx, hipStreamWaitEvent(c.sa, L.ev[30], 0));       // the ragged blocks are walked
            rc_round.push_back(r);
            hipLaunchKernelGGL((k_traceback<kWordsA, kTWordsA, kColsA, kOpsA, false, 5, kOcaBlockSize, false, 4>), dim3((gA + 3) / 4), dim3(256), 0, c.sa, itA, bound, d_nA, c.cap,
                               (const u64*)c.fragA, (const char*)slabsA, kSlabA, (const BlockResult*)c.resA, c.opsA, c.tasks, X.tail_match_len,
                               (i32*)nullptr, X.d_err, next, fl_all, 0u, (const WalkOut*)wo);
            NECAT_CHECK_LAUNCH(ctx, "k_traceback<A, rc>");
            if (!g_rc_ragged || wide_possible) NECAT_HIP(ctx, hipStreamWaitEvent(c.sa, L.ev[25], 0));          // the round is over when both chains are
        } else
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
    const int nlanes = overlap && n_batches > 1 ? 2 : 1;
    const u32 groups = cap / 64 + 1;
    int rc;
    // candidate-wide arrays
    const uint64_t n_groups_max = n;
    const size_t cand_bytes = n * sizeof(necat_candidate) + 2 * n * sizeof(necat_m4) + ((n + 63) & ~63ULL) + (n_groups_max + 1) * 8 + 1024;
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_EXT_CAND], cand_bytes))) return rc;
    char* cb = (char*)ctx->scratch[SC_EXT_CAND].p;
    necat_candidate* d_cands = (necat_candidate*)cb; cb += n * sizeof(necat_candidate);
    necat_m4* d_m4 = (necat_m4*)cb; cb += n * sizeof(necat_m4);
    necat_m4* d_out = (necat_m4*)cb; cb += n * sizeof(necat_m4);
    u64* d_goff = (u64*)cb; cb += (n_groups_max + 1) * 8;
    u32* d_outcnt = (u32*)cb; cb += 256;          // [0..1] output counter, [2..17] list counters (4 buffers x 4) of lane 0, [34..49] of lane 1
    int* d_err = (int*)cb; cb += 64;
    u8* d_ok = (u8*)cb;
    NECAT_HIP(ctx, hipMemcpyAsync(d_cands, dev ? dev->d : cands, n * sizeof(necat_candidate), dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    NECAT_HIP(ctx, hipMemsetAsync(d_outcnt, 0, 320, s));
    auto cleanup = [&]() {};
    ExtLane lane[2];
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
    Batch kb[2];
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
    if (nlanes == 2) {
        // ---- two lanes: batch i + 1 starts on the free lane once batch i is in its tail (BatchRun::tail); ONE host thread turns both round loops,
        // whichever has its next list sizes published (BatchRun::ready) - the host still never waits for the device inside a loop
        struct LaneRun { std::unique_ptr<BatchRun> run; int state = 0; int rc = NECAT_OK; };      // state: 0 free, 1 in its rounds, 2 draining
        LaneRun lr[2];
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
                for (int l = 0; l < 2; ++l) if (lr[l].state == 0) {
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
            for (int l = 0; l < 2 && !err; ++l) {
                LaneRun& R = lr[l];
                if (R.state == 1 && R.run->ready()) { R.rc = R.run->step(); progressed = true; if (R.run->over) R.state = 2; }
                if (R.state == 2 && (R.rc || R.run->drained())) {
                    if (!(err = R.run->finish(R.rc))) {
                        hipLaunchKernelGGL(k_ext_result, dim3(grid_for(kb[l].n, 256)), dim3(256), 0, kb[l].sa, (const ExtTask*)kb[l].tasks, kb[l].n, (const necat_candidate*)d_cands,
                                           opt->align_size_cutoff, d_m4, d_ok, rm ? 1 : 0);
                        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(kb[l].sa) != hipSuccess) err = set_err(ctx, NECAT_ERR_DEVICE, "k_ext_result failed");
                    }
                    R.run.reset(); R.state = 0; ++done; progressed = true;
                }
            }
            if (progressed) { idle = 0; t_idle = wall_ms(); continue; }
            if ((++idle & 0xfffff) == 0) {
                // a failed kernel never publishes: look at the streams instead of spinning forever
                for (int l = 0; l < 2 && !err; ++l) if (lr[l].state == 1) {
                    const hipError_t q = hipStreamQuery(kb[l].sa);
                    if (q != hipSuccess && q != hipErrorNotReady) err = set_err(ctx, NECAT_ERR_DEVICE, "extension rounds (lane %d) failed: %s", l, hipGetErrorString(q));
                }
                if (!err && wall_ms() - t_idle > 120e3) err = set_err(ctx, NECAT_ERR_DEVICE, "extension rounds: no progress for 120 s");
            }
        }
        if (err) {
            for (LaneRun& R : lr) if (R.state) { (void)R.run->finish(err); R.run.reset(); }       // nothing of a lane is in flight when its buffers are handed on
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
            hipLaunchKernelGGL((k_ext_frag<kAsmWordsA, kAsmTWordsA>), dim3(grid_for((u64)gA * 64 * (kAsmWordsA + kAsmTWordsA), 256)), dim3(256), 0, s,
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
            } else
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
        }
        // ---- list B: a plain list of nB items
        if (nB) {
            const u32 gB = (nB + 63) / 64;
            hipLaunchKernelGGL((k_ext_frag<kAsmWords, kAsmTWords>), dim3(grid_for((u64)gB * 64 * (kAsmWords + kAsmTWords), 256)), dim3(256), 0, sB,
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
            } else
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
}

// ------------------------------------------------------------------------------------------ oc2asmpm: votes and chained ranges on the device (asm_plan.h)

int necat_asm_plan_batch(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads, int read_start_id, int ref_start_id,
                         const necat_map_options* opt, necat_asm_plan** out, uint64_t** first)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ix || !ref || !reads || !opt || !out || !first) return NECAT_ERR_ARG;
    *out = nullptr; *first = nullptr;
    if (opt->kmer_size != ix->k) return set_err(ctx, NECAT_ERR_ARG, "index was built for k=%d, options say %d", ix->k, opt->kmer_size);
    if (opt->scan_window < 1 || opt->num_candidates < 1 || opt->num_candidates > 65536) return set_err(ctx, NECAT_ERR_ARG, "scan_window / num_candidates out of range");
    if (ref->nbases >= (1ULL << 31) || reads->nbases >= (1ULL << 31)) return set_err(ctx, NECAT_ERR_ARG, "volume too large for 32-bit offsets (asm_pm_common.c keeps them in int)");
    NECAT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const u32 nreads = (u32)reads->nseq;
    const int NE = opt->num_candidates;
    uint64_t* fo = (uint64_t*)result_alloc(((size_t)nreads + 1) * 8);
    if (!fo) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
    auto fail = [&](int rc) { necat_free(fo); return rc; };
    if (nreads == 0) { fo[0] = 0; *first = fo; *out = (necat_asm_plan*)result_alloc(sizeof(necat_asm_plan)); return NECAT_OK; }
    const DevVolume dref = dev_view(ref), drd = dev_view(reads);
    int rc;
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_prev = t_begin;
    auto tick = [&](const char* what) {          // (host clock between the calls' own synchronisation points; NECAT_TRACE=4 - it must not add any: the chunks overlap)
        if (!(g_trace & 4)) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[necat] asm plan %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // ---- hit counts per read-strand (k_seed_hits with z = BC), the table words kept
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_MISC], (size_t)nreads * 8 + 128)) ||
        (rc = buf_ensure(ctx, ctx->scratch[SC_SEED_KST], 2 * (reads->nbases / (u64)opt->scan_window + nreads + 2) * 8))) return fail(rc);
    u32* d_hits = (u32*)ctx->scratch[SC_MISC].p;
    int* d_err = (int*)((char*)ctx->scratch[SC_MISC].p + (((size_t)nreads * 8 + 63) & ~(size_t)63));
    u64* d_kst = (u64*)ctx->scratch[SC_SEED_KST].p;
    hipLaunchKernelGGL(k_seed_hits, dim3(grid_for((u64)nreads * 64, 256)), dim3(256), 0, s, drd, index_view(ix), opt->kmer_size, opt->scan_window, 0u, nreads, d_hits, d_kst);
    if (hipGetLastError() != hipSuccess) return fail(set_err(ctx, NECAT_ERR_DEVICE, "k_seed_hits launch failed"));
    std::vector<u32> hits((size_t)nreads * 2);
    if (hipMemsetAsync(d_err, 0, 4, s) != hipSuccess || hipMemcpyAsync(hits.data(), d_hits, (size_t)nreads * 8, hipMemcpyDeviceToHost, s) != hipSuccess)
        return fail(set_err(ctx, NECAT_ERR_DEVICE, "asm plan: hit counts"));
    // ---- per (subject, strand): occurrences of the sampled 10-mers (beside the copy above)
    u64 ref_lmax = 0;
    for (u64 q = 0; q < ref->nseq; ++q) ref_lmax = std::max(ref_lmax, ref->h_seq_off[q + 1] - ref->h_seq_off[q]);
    u32 cap_max = 64; while (cap_max < 2 * (ref_lmax / kAsmRangeW + 1)) cap_max <<= 1;
    const u32 occ_waves = (u32)std::max<u64>(1, std::min<u64>(std::min<u64>(2 * ref->nseq, 4096), ((u64)1 << 30) / ((u64)cap_max * 8)));
    const size_t occ_bytes = 2 * (ref->nbases / kAsmRangeW + ref->nseq + 2);
    if ((rc = buf_ensure(ctx, ctx->scratch[SC_ASM_OCC], occ_bytes)) || (rc = buf_ensure(ctx, ctx->scratch[SC_ASM_TAB], (size_t)occ_waves * cap_max * 8))) return fail(rc);
    u8* d_occ = (u8*)ctx->scratch[SC_ASM_OCC].p;
    if (hipMemsetAsync(ctx->scratch[SC_ASM_TAB].p, 0, (size_t)occ_waves * cap_max * 8, s) != hipSuccess) return fail(set_err(ctx, NECAT_ERR_DEVICE, "asm plan: memset"));
    if (ref->nseq) hipLaunchKernelGGL(k_asm_subj_occ, dim3(occ_waves), dim3(64), 0, s, dref, (u32*)ctx->scratch[SC_ASM_TAB].p, cap_max, d_occ);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return fail(set_err(ctx, NECAT_ERR_DEVICE, "k_asm_subj_occ failed: %s", hipGetErrorString(hipGetLastError())));
    tick("hits + subject occurrences");
    // ---- reads in descending work order, chunks bounded by a scratch budget (the pool of 384-byte blocks is sized by the hit counts)
    std::vector<u32> order(nreads);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return std::max(hits[2 * (size_t)a], hits[2 * (size_t)a + 1]) > std::max(hits[2 * (size_t)b], hits[2 * (size_t)b + 1]); });
    // (the pool of 384-byte blocks is sized by the hit counts, an upper bound several times the blocks really touched: the budget is what keeps a
    // chunk inside HBM.  A chunk's vote kernels are as long as the walk of its heaviest read, so the chunks run on TWO arena sets and two streams:
    // chunk i + 1's vote kernels are in flight while chunk i's tail finishes and its range stage runs.  16 M blocks = 6 GB per set by default - a
    // short-lived process pays for the device memory it maps (the first 30 GB of arenas of a process on a fresh box took 0.9 s,
    // profiles/NOTES_r04.md 4) - and the two sets' pools + candidate lists (the per-block bytes below: both scale with the budget; the hash tables, the
    // selection and read-index arenas are small beside them) together never more than 40 % of the memory that is free now)
    u64 budget_blocks = getenv("NECAT_ASM_VOTE_BUDGET") ? std::max<u64>(1024, strtoull(getenv("NECAT_ASM_VOTE_BUDGET"), nullptr, 10)) : (u64)16 << 20;
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr) budget_blocks = std::max<u64>(1 << 16, std::min<u64>(budget_blocks, (u64)(fr * 0.4) / (2 * (sizeof(VBlock) + sizeof(VoteCand)))));
    }
    static const u64 budget_seeds = getenv("NECAT_ASM_SEED_BUDGET") ? std::max<u64>(1024, strtoull(getenv("NECAT_ASM_SEED_BUDGET"), nullptr, 10)) : (u64)32 << 20;
    static const bool overlap = !getenv("NECAT_ASM_NO_OVERLAP");            // (A/B: one arena set, one stream, chunk after chunk)
    VoteParams P; P.k = opt->kmer_size; P.bc = opt->scan_window; P.read_start_id = read_start_id; P.ref_start_id = ref_start_id; P.num_extended = NE;
    std::vector<std::vector<necat_asm_plan>> per_read(nreads);
    u64 tot_pairs = 0, tot_seeds = 0, tot_plans = 0;
    // the chunks, and every per-chunk arena sized once for the largest of them (a grow-only arena that grows chunk by chunk is freed and
    // allocated again each time)
    auto both = [&](u32 r) { return (u64)hits[2 * (size_t)r] + hits[2 * (size_t)r + 1] + 2; };
    std::vector<u32> chunk_end;
    static const ScratchId kSet[2][7] = {{SC_ASM_VMETA, SC_ASM_VHT, SC_ASM_VPOOL, SC_ASM_VOUT, SC_ASM_SEL, SC_ASM_RIDX, SC_ASM_RNEXT},
                                         {SC_ASM_VMETA2, SC_ASM_VHT2, SC_ASM_VPOOL2, SC_ASM_VOUT2, SC_ASM_SEL2, SC_ASM_RIDX2, SC_ASM_RNEXT2}};
    {
        u64 mx_n = 0, mx_ht = 0, mx_pool = 0, mx_tab = 0, mx_next = 0;
        for (u32 p0 = 0; p0 < nreads;) {
            u64 acc = 0, ht = 0, tab = 0, nx = 0; u32 h1 = p0;
            while (h1 < nreads && (h1 == p0 || acc + both(order[h1]) <= budget_blocks)) {
                const u32 r = order[h1];
                acc += both(r);
                for (int st = 0; st < 2; ++st) { const u64 H = std::max<u64>(1, hits[2 * (size_t)r + st]); u64 cap = 4; while (cap < 2 * H) cap <<= 1; ht += cap; }
                const u64 L = reads->h_seq_off[r + 1] - reads->h_seq_off[r];
                tab += 2 * (L + L / 2 + 64); nx += L + 1;
                ++h1;
            }
            mx_n = std::max<u64>(mx_n, h1 - p0); mx_ht = std::max(mx_ht, ht); mx_pool = std::max(mx_pool, acc); mx_tab = std::max(mx_tab, tab); mx_next = std::max(mx_next, nx);
            chunk_end.push_back(h1);
            p0 = h1;
        }
        const size_t meta_bytes = (size_t)mx_n * (sizeof(VoteMeta) + sizeof(ReadIdxMeta) + 4 /* order */ + 8 /* nblk */ + 8 /* nstrand */ + 4 /* nplan */) + 512;
        for (int e = 0; e < ((overlap && chunk_end.size() > 1) ? 2 : 1); ++e)
            if ((rc = buf_ensure(ctx, ctx->scratch[kSet[e][0]], meta_bytes)) ||
                (rc = buf_ensure(ctx, ctx->scratch[kSet[e][1]], mx_ht * 8)) ||
                (rc = buf_ensure(ctx, ctx->scratch[kSet[e][2]], mx_pool * sizeof(VBlock))) ||
                (rc = buf_ensure(ctx, ctx->scratch[kSet[e][3]], mx_pool * sizeof(VoteCand))) ||
                (rc = buf_ensure(ctx, ctx->scratch[kSet[e][4]], (size_t)mx_n * NE * (sizeof(VoteCand) + sizeof(AsmPlanDev)))) ||
                (rc = buf_ensure(ctx, ctx->scratch[kSet[e][5]], mx_tab * 4)) ||
                (rc = buf_ensure(ctx, ctx->scratch[kSet[e][6]], mx_next * 4))) return fail(rc);
    }
    hipStream_t st2[2] = {s, s};
    if (overlap && chunk_end.size() > 1) {
        if (int rcs = ext_streams(ctx)) return fail(rcs);          // (the context's one place that makes streams: NECAT_SERIAL aliases and NECAT_STREAM_PRIO apply here too)
        st2[1] = ctx->stream_b;
    }
    tick("chunk plan + arenas");
    // what a chunk leaves on the device between its two halves
    struct Chunk { u32 pos = 0, n = 0; VoteMeta* d_meta = nullptr; ReadIdxMeta* d_rmeta = nullptr; u32* d_order = nullptr; i32 *d_nblk = nullptr, *d_nstrand = nullptr, *d_nplan = nullptr;
                   VoteArenas A; VoteCand* d_sel = nullptr; AsmPlanDev* d_plan = nullptr; u32 *d_tabs = nullptr, *d_next = nullptr; };
    Chunk chunks2[2];
    // ---- first half of a chunk: vote of both strands, the per-read cut, the reads' 10-mer tables - launched, not waited for
    auto launch_vote = [&](size_t ci, int e) -> int {
        hipStream_t sv = st2[e];
        Chunk& C = chunks2[e];
        C.pos = ci ? chunk_end[ci - 1] : 0u; C.n = chunk_end[ci] - C.pos;
        const u32 n = C.n, pos = C.pos;
        std::vector<VoteMeta> meta(n);
        std::vector<ReadIdxMeta> rmeta(n);
        u64 ht_tot = 0, pool_tot = 0, tab_tot = 0, next_tot = 0;
        for (u32 i = 0; i < n; ++i) {
            const u32 r = order[pos + i];
            for (int st = 0; st < 2; ++st) {
                const u64 H = std::max<u64>(1, hits[2 * (size_t)r + st]);
                u64 cap = 4; while (cap < 2 * H) cap <<= 1;
                meta[i].ht_off[st] = ht_tot; meta[i].ht_mask[st] = (u32)(cap - 1); ht_tot += cap;
                meta[i].pool_off[st] = pool_tot; meta[i].pool_cap[st] = (u32)H; pool_tot += H;
            }
            const u64 L = reads->h_seq_off[r + 1] - reads->h_seq_off[r];
            const u64 cap = L + L / 2 + 64;
            rmeta[i].tab_off = tab_tot; rmeta[i].next_off = next_tot; rmeta[i].cap = (u32)cap; rmeta[i]._pad = 0;
            tab_tot += 2 * cap; next_tot += L + 1;
        }
        char* mb = (char*)ctx->scratch[kSet[e][0]].p;
        auto carve = [&](size_t bytes) { char* q = mb; mb += (bytes + 63) & ~(size_t)63; return q; };
        C.d_meta = (VoteMeta*)carve(n * sizeof(VoteMeta));
        C.d_rmeta = (ReadIdxMeta*)carve(n * sizeof(ReadIdxMeta));
        C.d_order = (u32*)carve((size_t)n * 4);
        C.d_nblk = (i32*)carve((size_t)n * 8);
        C.d_nstrand = (i32*)carve((size_t)n * 8);
        C.d_nplan = (i32*)carve((size_t)n * 4);
        C.A.ht = (u64*)ctx->scratch[kSet[e][1]].p; C.A.pool = (VBlock*)ctx->scratch[kSet[e][2]].p; C.A.out = (VoteCand*)ctx->scratch[kSet[e][3]].p;
        C.d_sel = (VoteCand*)ctx->scratch[kSet[e][4]].p;
        C.d_plan = (AsmPlanDev*)((char*)ctx->scratch[kSet[e][4]].p + (size_t)n * NE * sizeof(VoteCand));
        C.d_tabs = (u32*)ctx->scratch[kSet[e][5]].p; C.d_next = (u32*)ctx->scratch[kSet[e][6]].p;
        if (hipMemcpyAsync(C.d_meta, meta.data(), n * sizeof(VoteMeta), hipMemcpyHostToDevice, sv) != hipSuccess ||
            hipMemcpyAsync(C.d_rmeta, rmeta.data(), n * sizeof(ReadIdxMeta), hipMemcpyHostToDevice, sv) != hipSuccess ||
            hipMemcpyAsync(C.d_order, order.data() + pos, (size_t)n * 4, hipMemcpyHostToDevice, sv) != hipSuccess ||
            hipMemsetAsync(C.A.ht, 0xFF, ht_tot * 8, sv) != hipSuccess ||
            hipMemsetAsync(C.d_tabs, 0, tab_tot * 4, sv) != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: chunk upload");
        hipLaunchKernelGGL(k_asm_vote_collect, dim3(2 * n), dim3(64), 0, sv, dref, drd, index_view(ix), (const u64*)ix->offset_list, P, (const u32*)C.d_order, (const VoteMeta*)C.d_meta, n, C.A,
                           C.d_nblk, d_err, (const u64*)d_kst);
        hipLaunchKernelGGL(k_asm_vote_eval, dim3(2 * n), dim3(64), 0, sv, dref, drd, P, (const u32*)C.d_order, (const VoteMeta*)C.d_meta, n, C.A, (const i32*)C.d_nblk, C.d_nstrand);
        hipLaunchKernelGGL(k_asm_select, dim3(n), dim3(64), 0, sv, P, (const VoteMeta*)C.d_meta, n, C.A, (const i32*)C.d_nstrand, C.d_sel, C.d_plan, C.d_nplan);
        hipLaunchKernelGGL(k_asm_read_index, dim3(n), dim3(64), 0, sv, drd, (const u32*)C.d_order, (const ReadIdxMeta*)C.d_rmeta, n, C.d_tabs, C.d_next);
        if (hipGetLastError() != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: vote kernels launch failed");
        return NECAT_OK;
    };
    // ---- second half: the planned pairs' match counts, matches, chains; the chunk's plan to the host
    auto finish_chunk = [&](int e) -> int {
        hipStream_t sv = st2[e];
        Chunk& C = chunks2[e];
        const u32 n = C.n, pos = C.pos;
        std::vector<i32> nplan(n);
        int herr = 0;
        if (hipMemcpyAsync(nplan.data(), C.d_nplan, (size_t)n * 4, hipMemcpyDeviceToHost, sv) != hipSuccess || hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, sv) != hipSuccess ||
            hipStreamSynchronize(sv) != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: vote kernels failed: %s", hipGetErrorString(hipGetLastError()));
        if (herr) return set_err(ctx, NECAT_ERR_CAPACITY, "asm plan: vote scratch overflow (code %d)", herr);
        tick("vote + select + read index");
        if (const char* dump = getenv("NECAT_ASM_DUMP_VOTES")) {
            // tests/host_core/check_asm_plan.cpp: per read {read id, candidates of both strands, kept}, then the ranked candidates (6 ints each)
            std::vector<VoteCand> hsel((size_t)n * NE);
            std::vector<i32> hns((size_t)n * 2);
            if (hipMemcpy(hsel.data(), C.d_sel, hsel.size() * sizeof(VoteCand), hipMemcpyDeviceToHost) == hipSuccess &&
                hipMemcpy(hns.data(), C.d_nstrand, hns.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
                if (FILE* f = fopen(dump, "ab")) {
                    for (u32 i = 0; i < n; ++i) {
                        const i32 tot = hns[2 * (size_t)i] + hns[2 * (size_t)i + 1], kept = std::min<i32>(tot, NE);
                        const i32 hdr[3] = {(i32)order[pos + i], tot, kept};
                        fwrite(hdr, 4, 3, f);
                        fwrite(hsel.data() + (size_t)i * NE, sizeof(VoteCand), (size_t)kept, f);
                    }
                    fclose(f);
                }
            }
        }
        std::vector<PairMeta> pairs;
        for (u32 i = 0; i < n; ++i) for (i32 q = 0; q < nplan[i]; ++q) { PairMeta pm; pm.read_i = i; pm.slot = (u32)q; pm.seed_off = 0; pairs.push_back(pm); }
        const u32 np = (u32)pairs.size();
        tot_pairs += np;
        std::vector<AsmPlanDev> hplan;
        int rc2;
        if (np) {
            if ((rc2 = buf_ensure(ctx, ctx->scratch[SC_ASM_PAIRS], (size_t)np * (sizeof(PairMeta) + 8) + 256))) return rc2;
            PairMeta* d_pairs = (PairMeta*)ctx->scratch[SC_ASM_PAIRS].p;
            u32* d_counts = (u32*)((char*)d_pairs + (((size_t)np * sizeof(PairMeta) + 63) & ~(size_t)63));
            u32* d_nmem = d_counts + np;
            if (hipMemcpyAsync(d_pairs, pairs.data(), (size_t)np * sizeof(PairMeta), hipMemcpyHostToDevice, sv) != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: pair upload");
            hipLaunchKernelGGL(k_asm_seeds<false>, dim3(np), dim3(64), 0, sv, dref, drd, (const u32*)C.d_order, (const ReadIdxMeta*)C.d_rmeta, (const u32*)C.d_tabs, (const u32*)C.d_next, (const u8*)d_occ,
                               (const AsmPlanDev*)C.d_plan, NE, (const PairMeta*)d_pairs, np, d_counts, (AsmSeed*)nullptr, (AsmMem*)nullptr, (AsmMem*)nullptr, (u32*)nullptr);
            std::vector<u32> counts(np);
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(counts.data(), d_counts, (size_t)np * 4, hipMemcpyDeviceToHost, sv) != hipSuccess || hipStreamSynchronize(sv) != hipSuccess)
                return set_err(ctx, NECAT_ERR_DEVICE, "k_asm_seeds<count> failed: %s", hipGetErrorString(hipGetLastError()));
            tick("match counts");
            const size_t per_seed = sizeof(AsmSeed) + 2 * sizeof(AsmMem) + 16;
            {   // the arena once per chunk, for its largest batch (+ a quarter: the next chunk's is about as large)
                u64 mx = 0, so = 0;
                for (u32 b = 0; b < np; ++b) { if (so && so + counts[b] > budget_seeds) { mx = std::max(mx, so); so = 0; } so += counts[b]; }
                mx = std::max(mx, so);
                if (std::max<u64>(1, mx) * per_seed + 256 > ctx->scratch[SC_ASM_SEEDS].cap && (rc2 = buf_ensure(ctx, ctx->scratch[SC_ASM_SEEDS], (std::max<u64>(1, mx) + mx / 4) * per_seed + 256))) return rc2;
            }
            for (u32 b0 = 0; b0 < np;) {
                u64 so = 0; u32 b1 = b0;
                while (b1 < np && (b1 == b0 || so + counts[b1] <= budget_seeds)) { pairs[b1].seed_off = so; so += counts[b1]; ++b1; }
                tot_seeds += so;
                char* sb = (char*)ctx->scratch[SC_ASM_SEEDS].p;
                AsmSeed* d_seeds = (AsmSeed*)sb; sb += ((so * sizeof(AsmSeed)) + 63) & ~(size_t)63;
                AsmMem* d_mems = (AsmMem*)sb; sb += ((so * sizeof(AsmMem)) + 63) & ~(size_t)63;
                AsmMem* d_tmp = (AsmMem*)sb; sb += ((so * sizeof(AsmMem)) + 63) & ~(size_t)63;
                i32* d_chain = (i32*)sb;
                const u32 nb = b1 - b0;
                if (hipMemcpyAsync(d_pairs + b0, pairs.data() + b0, (size_t)nb * sizeof(PairMeta), hipMemcpyHostToDevice, sv) != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: pair upload");
                hipLaunchKernelGGL(k_asm_seeds<true>, dim3(nb), dim3(64), 0, sv, dref, drd, (const u32*)C.d_order, (const ReadIdxMeta*)C.d_rmeta, (const u32*)C.d_tabs, (const u32*)C.d_next, (const u8*)d_occ,
                                   (const AsmPlanDev*)C.d_plan, NE, (const PairMeta*)(d_pairs + b0), nb, (u32*)nullptr, d_seeds, d_mems, d_tmp, d_nmem + b0);
                hipLaunchKernelGGL(k_asm_chain, dim3(nb), dim3(64), 0, sv, (const PairMeta*)(d_pairs + b0), nb, (const AsmMem*)d_mems, (const u32*)(d_nmem + b0), d_chain, C.d_plan, NE);
                if (hipGetLastError() != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: range kernels launch failed");
                if (b1 < np && hipStreamSynchronize(sv) != hipSuccess) return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: range kernels failed: %s", hipGetErrorString(hipGetLastError()));
                b0 = b1;
            }
            hplan.resize((size_t)n * NE);
            if (hipMemcpyAsync(hplan.data(), C.d_plan, (size_t)n * NE * sizeof(AsmPlanDev), hipMemcpyDeviceToHost, sv) != hipSuccess || hipStreamSynchronize(sv) != hipSuccess)
                return set_err(ctx, NECAT_ERR_DEVICE, "asm plan: range kernels failed: %s", hipGetErrorString(hipGetLastError()));
            tick("matches + chains");
        }
        for (u32 i = 0; i < n; ++i) {
            std::vector<necat_asm_plan>& dst = per_read[order[pos + i]];
            dst.resize((size_t)nplan[i]);
            for (i32 q = 0; q < nplan[i]; ++q) {
                const AsmPlanDev& en = hplan[(size_t)i * NE + (size_t)q];
                necat_asm_plan& o = dst[(size_t)q];
                o.qid = (int32_t)order[pos + i] + read_start_id; o.sid = en.sid + ref_start_id; o.sdir = en.sdir; o.qoff = en.qoff; o.soff = en.soff; o.score = en.score; o.ssize = en.ssize;
            }
            tot_plans += (u64)nplan[i];
        }
        return NECAT_OK;
    };
    {
        const size_t nch = chunk_end.size();
        const int two = (overlap && nch > 1) ? 1 : 0;
        auto drain = [&]() { (void)hipStreamSynchronize(st2[0]); (void)hipStreamSynchronize(st2[1]); };       // nothing in flight when an error returns
        if ((rc = launch_vote(0, 0))) { drain(); return fail(rc); }
        for (size_t ci = 0; ci < nch; ++ci) {
            const int e = two ? (int)(ci & 1) : 0;
            if (two && ci + 1 < nch && (rc = launch_vote(ci + 1, e ^ 1))) { drain(); return fail(rc); }
            if ((rc = finish_chunk(e))) { drain(); return fail(rc); }
            if (!two && ci + 1 < nch && (rc = launch_vote(ci + 1, 0))) { drain(); return fail(rc); }
        }
    }
    necat_asm_plan* res = (necat_asm_plan*)result_alloc(std::max<u64>(1, tot_plans) * sizeof(necat_asm_plan));
    if (!res) return fail(set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"));
    u64 at = 0;
    for (u32 r = 0; r < nreads; ++r) { fo[r] = at; for (const necat_asm_plan& e : per_read[r]) res[at++] = e; }
    fo[nreads] = at;
    if (g_trace & 2) fprintf(stderr, "[necat] asm plan: %u reads, %lu planned pairs, %lu matches, %.2f ms\n", nreads, (unsigned long)tot_pairs, (unsigned long)tot_seeds,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    *out = res; *first = fo;
    return NECAT_OK;
}

// ------------------------------------------------------------------------------------------ reads against a reference (oc2rm_worker)

int necat_map_reference(necat_ctx* ctx, const necat_index* ix, const necat_volume* ref, const necat_volume* reads,
                        int read_start_id, int ref_start_id, const necat_map_options* opt,
                        necat_m4** out, uint64_t* n_out, uint64_t* n_candidates, uint64_t* n_rescued)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ix || !ref || !reads || !opt || !out || !n_out) return NECAT_ERR_ARG;
    *out = nullptr; *n_out = 0;
    if (n_candidates) *n_candidates = 0;
    if (n_rescued) *n_rescued = 0;
    necat_map_options o = *opt;
    o.job = 1;                                   // rm_worker.c:251-252: sorted, cut to num_candidates
    DevCands dev;
    int rc = find_impl(ctx, ix, ref, reads, read_start_id, ref_start_id, 0 /* pairwise = FALSE, rm_worker.c:231 */, &o, nullptr, nullptr, &dev);
    if (rc) return rc;
    if (n_candidates) *n_candidates = dev.n;
    if (dev.n == 0) { *out = (necat_m4*)result_alloc(sizeof(necat_m4)); return *out ? NECAT_OK : set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"); }
    RmOut ro;
    necat_m4* unused = nullptr; uint64_t unused_n = 0;
    if ((rc = extend_impl(ctx, ref, reads, read_start_id, ref_start_id, nullptr, dev.n, &o, 1 /* ONC_TAIL_MATCH_LEN_SHORT, rm_worker.c:92 */,
                          &unused, &unused_n, nullptr, &dev, nullptr, &ro))) return rc;
    const double w0 = wall_ms();
    // the bases come back to the host only if some candidate needs the rescue pair
    const uint64_t ng = ro.group_off.empty() ? 0 : ro.group_off.size() - 1;
    bool any = false;
    for (uint64_t i = 0; i < dev.n && !any; ++i) any = ro.ok[i] && rm::needs_rescue(ro.cands[i], ro.m4[i]);
    std::vector<u64> w_reads, w_ref;
    if (any) {
        w_reads.resize((reads->nbases + 31) / 32 + 1); w_ref.resize((ref->nbases + 31) / 32 + 1);
        NECAT_HIP(ctx, hipMemcpy(w_reads.data(), reads->bases, (w_reads.size() - 1) * 8, hipMemcpyDeviceToHost));
        NECAT_HIP(ctx, hipMemcpy(w_ref.data(), ref->bases, (w_ref.size() - 1) * 8, hipMemcpyDeviceToHost));
    }
    rm::Words hr, hf;
    hr.w = w_reads.data(); hr.seq_off = reads->h_seq_off.data();
    hf.w = w_ref.data(); hf.seq_off = ref->h_seq_off.data();
    const rescue::DalignSpec dspec = rescue::spec_for_error(o.error);
    // groups (reads) are dealt out in runs of 16; every run's records are kept apart and joined in read order
    const uint64_t run = 16, nruns = (ng + run - 1) / run;
    std::vector<std::vector<necat_m4>> parts(nruns);
    std::atomic<uint64_t> next(0), tried(0), rescued(0);
    auto work = [&]() {
        rm::Worker wk(dspec, o.error);
        for (;;) {
            const uint64_t r = next.fetch_add(1);
            if (r >= nruns) break;
            for (uint64_t g = r * run; g < std::min(ng, (r + 1) * run); ++g)
                wk.replay(ro.cands.data(), ro.m4.data(), ro.ok.data(), ro.group_off[g], ro.group_off[g + 1], hr, hf, read_start_id, ref_start_id,
                          o.align_size_cutoff, parts[r]);
        }
        tried += wk.n_rescue_tried; rescued += wk.n_rescued;
    };
    unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 32u));
    nt = (unsigned)std::min<uint64_t>(nt, std::max<uint64_t>(1, nruns));
    std::vector<std::thread> th;
    for (unsigned x = 0; x + 1 < nt; ++x) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
    uint64_t total = 0;
    for (auto& p : parts) total += p.size();
    necat_m4* res = (necat_m4*)result_alloc(std::max<uint64_t>(1, total) * sizeof(necat_m4));
    if (!res) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
    uint64_t at = 0;
    for (auto& p : parts) { if (!p.empty()) memcpy(res + at, p.data(), p.size() * sizeof(necat_m4)); at += p.size(); }
    if (n_rescued) *n_rescued = rescued.load();
    if (g_trace & 2) fprintf(stderr, "[necat] map_reference host: %.2f ms, %lu candidates, %lu rescue attempts, %lu rescued, %lu records\n", wall_ms() - w0,
                             (unsigned long)dev.n, (unsigned long)tried.load(), (unsigned long)rescued.load(), (unsigned long)total);
    *out = res; *n_out = total;
    return NECAT_OK;
}

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
        if (!all_ok) {
            if (c->nccl && c->p_CommDestroy) { (void)c->p_CommDestroy(c->nccl); c->nccl = nullptr; }
            if (!was_auto) { const int e = ok ? set_err(ctx, NECAT_ERR_COMM, "RCCL could not be initialised on another rank") : NECAT_ERR_COMM; delete c; return e; }
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

int necat_onc_align_batch(necat_ctx* ctx, const necat_volume* ref, const necat_volume* reads, int read_start_id, int ref_start_id,
                          const necat_candidate* cands, uint64_t n, const necat_map_options* opt, int tail_match_len,
                          necat_alignment** aln, uint8_t** ops, uint64_t** ops_off)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !ref || !reads || !opt || !aln || !ops || !ops_off || (n && !cands)) return NECAT_ERR_ARG;
    *aln = nullptr; *ops = nullptr; *ops_off = nullptr;
    AlignOut ao;
    ao.aln = (necat_alignment*)result_alloc(std::max<uint64_t>(1, n) * sizeof(necat_alignment));
    if (!ao.aln) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
    ao.off.assign(n + 1, 0);
    if (n) {
        const int rc = extend_impl(ctx, ref, reads, read_start_id, ref_start_id, cands, n, opt, tail_match_len, nullptr, nullptr, &ao);
        if (rc) { necat_free(ao.aln); for (auto& pr : ao.parts) necat_free(pr.first); return rc; }
    }
    uint64_t* f = (uint64_t*)result_alloc((n + 1) * 8);
    uint8_t* o = nullptr;
    if (ao.parts.size() == 1) { o = ao.parts[0].first; ao.parts.clear(); }       // the usual case: one batch, no copy
    else {
        o = (uint8_t*)result_alloc(std::max<uint64_t>(1, ao.total));
        uint64_t at = 0;
        if (o) for (auto& pr : ao.parts) { memcpy(o + at, pr.first, pr.second); at += pr.second; }
        for (auto& pr : ao.parts) necat_free(pr.first);
        ao.parts.clear();
    }
    if (!o || !f) { necat_free(ao.aln); necat_free(o); necat_free(f); return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"); }
    memcpy(f, ao.off.data(), (n + 1) * 8);
    *aln = ao.aln; *ops = o; *ops_off = f;
    return NECAT_OK;
}

int necat_gapped_strings(const uint8_t* ops, uint64_t n, const uint8_t* qseq, uint64_t qsize, uint64_t qoff,
                         const uint8_t* tseq, uint64_t tsize, uint64_t toff, char* query_align, char* target_align)
{
    if ((n && (!ops || !query_align || !target_align)) || !qseq || !tseq) return NECAT_ERR_ARG;
    static const char dec[5] = {'A', 'C', 'G', 'T', '-'};      // DecodeDNA / GAP_CHAR (common/ontcns_defs.h:36-39)
    uint64_t q = qoff, t = toff;
    for (uint64_t i = 0; i < n; ++i) {
        const int op = (ops[i >> 2] >> ((i & 3) * 2)) & 3;
        if ((op != 2 && q >= qsize) || (op != 1 && t >= tsize)) return NECAT_ERR_ARG;
        query_align[i] = op == 2 ? '-' : dec[qseq[q] & 3];
        target_align[i] = op == 1 ? '-' : dec[tseq[t] & 3];
        q += op != 2; t += op != 1;
    }
    return NECAT_OK;
}

// ------------------------------------------------------------------------------------------ consensus stage: the extension loop

void necat_cns_default_options(necat_cns_options* o)
{   // consensus/cns_options.c:10-22
    o->min_align_size = 400; o->min_cov = 4; o->max_cov = 12; o->error = 0.5; o->mapping_ratio = 0.8; o->use_fixed_ident_cutoff = 0;
    o->rescue_long_indels = 0;
}

int necat_cns_load_partition(necat_ctx* ctx, const necat_volume* reads, const void* packed, uint64_t n,
                             necat_candidate** cands, uint64_t** tmpl_off, uint64_t** n_all, uint64_t* n_templates)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !reads || (n && !packed) || !cands || !tmpl_off || !n_all || !n_templates) return NECAT_ERR_ARG;
    *cands = nullptr; *tmpl_off = nullptr; *n_all = nullptr; *n_templates = 0;
    std::vector<cns::Packed> recs(n);
    if (n) memcpy(recs.data(), packed, n * sizeof(cns::Packed));
    std::vector<necat_candidate> c; std::vector<uint64_t> off, na;
    const uint64_t bad = cns::load_partition(recs, reads->h_seq_off.data(), reads->nseq, c, off, na);
    if (bad) return set_err(ctx, NECAT_ERR_ARG, "candidate record %lu refers to a read outside the read set or has a range outside its reads", (unsigned long)(bad - 1));
    necat_candidate* oc = (necat_candidate*)malloc(std::max<size_t>(1, c.size()) * sizeof(necat_candidate));
    uint64_t* oo = (uint64_t*)malloc(off.size() * 8);
    uint64_t* on = (uint64_t*)malloc(std::max<size_t>(1, na.size()) * 8);
    if (!oc || !oo || !on) { free(oc); free(oo); free(on); return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed"); }
    if (!c.empty()) memcpy(oc, c.data(), c.size() * sizeof(necat_candidate));
    memcpy(oo, off.data(), off.size() * 8);
    if (!na.empty()) memcpy(on, na.data(), na.size() * 8);
    *cands = oc; *tmpl_off = oo; *n_all = on; *n_templates = na.size();
    return NECAT_OK;
}

void necat_cns_result_free(necat_cns_result* r)
{
    if (!r) return;
    for (uint32_t b = 0; b < r->n_ops_blocks; ++b) necat_free(r->ops[b]);
    free(r->ops); free(r->templates); free(r->overlaps); free(r->ranges);
    free(r);
}

int necat_cns_extension_batch(necat_ctx* ctx, const necat_volume* reads, const necat_candidate* cands, const uint64_t* tmpl_off,
                              const uint64_t* n_all, uint64_t n_templates, const necat_cns_options* opt, necat_cns_result** out)
{
    KnobScope knob_scope_(ctx);
    if (!ctx || !reads || !opt || !out || (n_templates && (!tmpl_off || !cands))) return NECAT_ERR_ARG;
    *out = nullptr;
    if (opt->max_cov < 1 || opt->max_cov > 60000 || opt->min_align_size < 0 || !(opt->error > 0.0 && opt->error <= 1.0))
        return set_err(ctx, NECAT_ERR_ARG, "consensus options out of range");
    const double w0 = wall_ms();
    std::vector<cns::Template> ts(n_templates);
    for (uint64_t t = 0; t < n_templates; ++t) {
        const uint64_t lo = tmpl_off[t], hi = tmpl_off[t + 1];
        if (hi < lo || hi - lo >= (1ULL << 31)) return set_err(ctx, NECAT_ERR_ARG, "template %lu: bad candidate range", (unsigned long)t);
        cns::Template& T = ts[t];
        T.c = cands + lo; T.c_base = lo; T.n = (uint32_t)(hi - lo); T.n_all = n_all ? (uint32_t)std::min<uint64_t>(n_all[t], 0xffffffffu) : T.n;
        for (uint64_t i = lo; i < hi; ++i) {
            const necat_candidate& c = cands[i];
            if (c.sid != cands[lo].sid || c.sdir != 0 || c.sid < 0 || (uint64_t)c.sid >= reads->nseq || c.qid < 0 || (uint64_t)c.qid >= reads->nseq ||
                c.ssize != reads->h_seq_off[c.sid + 1] - reads->h_seq_off[c.sid] || c.qsize != reads->h_seq_off[c.qid + 1] - reads->h_seq_off[c.qid] ||
                c.sbeg > c.send || c.send > c.ssize || c.qoff > c.qsize || c.soff > c.ssize || c.ssize >= (1ULL << 31) || c.qsize >= (1ULL << 31))
                return set_err(ctx, NECAT_ERR_ARG, "candidate %lu of template %lu is inconsistent (one forward subject per template, ranges inside the reads)",
                               (unsigned long)(i - lo), (unsigned long)t);
        }
        T.tsize = T.n ? (int)cands[lo].ssize : 0;
    }
    necat_map_options mo; necat_default_options(&mo);
    mo.error = opt->error; mo.align_size_cutoff = opt->min_align_size;
    std::vector<u8*> blocks;
    double device_ms = 0, align_wall = 0;
    // -r 1: the host pair (cns_rescue.h) on the candidates of a pass whose block-wise extension failed or fell short.  The reads
    // come back from the device once per call (2-bit words, base i in bits 2 (i & 31) of word i >> 5).
    std::vector<u64> h_words;
    const rescue::DalignSpec dspec = opt->rescue_long_indels ? rescue::spec_for_error(opt->error) : rescue::DalignSpec();
    uint64_t n_rescue_tried = 0, n_rescued = 0;
    double rescue_ms = 0;
    auto rescue_pass = [&](const necat_candidate* c, uint64_t m, cns::Aligned* res) -> int {
        const double r0 = wall_ms();
        std::vector<uint64_t> need;
        for (uint64_t i = 0; i < m; ++i) if (cns::extension_short(c[i], res[i].a)) need.push_back(i);
        if (need.empty()) return NECAT_OK;
        if (h_words.empty()) {
            h_words.resize((reads->nbases + 31) / 32 + 1);
            NECAT_HIP(ctx, hipMemcpy(h_words.data(), reads->bases, (h_words.size() - 1) * 8, hipMemcpyDeviceToHost));
        }
        struct Got { bool ok = false; necat_alignment a; std::vector<u8> packed; };
        std::vector<Got> got(need.size());
        std::atomic<size_t> next(0);
        auto work = [&]() {
            cns::Rescuer rs(dspec, opt->error);
            std::vector<u8> q, t;
            auto decode = [&](int32_t id, int rev, std::vector<u8>& dst) {
                const u64 b = reads->h_seq_off[id], n = reads->h_seq_off[id + 1] - b;
                dst.resize(n);
                if (!rev) for (u64 i = 0; i < n; ++i) dst[i] = (u8)((h_words[(b + i) >> 5] >> (((b + i) & 31) * 2)) & 3);
                else for (u64 i = 0; i < n; ++i) { const u64 g = b + n - 1 - i; dst[i] = (u8)(3 - ((h_words[g >> 5] >> ((g & 31) * 2)) & 3)); }
            };
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= need.size()) break;
                const necat_candidate& cc = c[need[k]];
                decode(cc.qid, cc.qdir, q); decode(cc.sid, 0, t);
                Got& g = got[k];
                g.a = res[need[k]].a;
                g.ok = rs.go(cc, q.data(), t.data(), opt->min_align_size, &g.a);
                if (!g.ok) continue;
                g.packed.assign((rs.cols.size() + 3) / 4, 0);
                for (size_t j = 0; j < rs.cols.size(); ++j) g.packed[j >> 2] |= (u8)(rs.cols[j] << (2 * (j & 3)));
            }
        };
        unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 32u));
        nt = (unsigned)std::min<size_t>(nt, need.size());
        std::vector<std::thread> th;
        for (unsigned x = 0; x + 1 < nt; ++x) th.emplace_back(work);
        work();
        for (auto& x : th) x.join();
        u64 bytes = 0;
        for (const Got& g : got) if (g.ok) bytes += (g.packed.size() + 7) & ~(u64)7;
        n_rescue_tried += need.size();
        if (bytes) {
            u8* blk = (u8*)result_alloc(bytes);
            if (!blk) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
            const u32 bi = (u32)blocks.size();
            blocks.push_back(blk);
            u64 at = 0;
            for (size_t k = 0; k < got.size(); ++k) {
                const Got& g = got[k];
                if (!g.ok) continue;
                memcpy(blk + at, g.packed.data(), g.packed.size());
                res[need[k]].a = g.a; res[need[k]].block = bi; res[need[k]].off = at;
                at += (g.packed.size() + 7) & ~(u64)7;
                ++n_rescued;
            }
        }
        rescue_ms += wall_ms() - r0;
        if (g_trace & 2) fprintf(stderr, "[necat] cns rescue: %zu of %lu candidates tried, %.2f ms\n", need.size(), (unsigned long)m, wall_ms() - r0);
        return NECAT_OK;
    };
    cns::AlignFn fn = [&](const necat_candidate* c, uint64_t m, cns::Aligned* res) -> int {
        const double a0 = wall_ms();
        AlignOut ao;
        ao.defer_copy = true;       // the loop only needs the coordinates to go on; the columns arrive while it does
        ao.aln = (necat_alignment*)result_alloc(std::max<uint64_t>(1, m) * sizeof(necat_alignment));
        if (!ao.aln) return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
        ao.off.assign(m + 1, 0);
        const int rc = extend_impl(ctx, reads, reads, 0, 0, c, m, &mo, 4 /* ONC_TAIL_MATCH_LEN_LONG, oc_aligner.h:42 */, nullptr, nullptr, &ao);
        if (rc) {
            if (ctx->stream_copy) (void)hipStreamSynchronize(ctx->stream_copy);
            ctx->copy_pending = false;
            necat_free(ao.aln); for (auto& pr : ao.parts) necat_free(pr.first); return rc;
        }
        device_ms += ctx->tm.extend_ms;
        // the columns stay where the device copied them: one block per batch of the pass
        size_t p = 0; u64 p_start = 0;
        const u32 b0 = (u32)blocks.size();
        for (auto& pr : ao.parts) blocks.push_back(pr.first);
        for (uint64_t i = 0; i < m; ++i) {
            res[i].a = ao.aln[i];
            const u64 at = ao.off[i];
            while (p < ao.parts.size() && at >= p_start + ao.parts[p].second && ao.off[i + 1] > at) { p_start += ao.parts[p].second; ++p; }
            res[i].block = b0 + (u32)std::min(p, ao.parts.empty() ? 0 : ao.parts.size() - 1);
            res[i].off = at - p_start;
        }
        necat_free(ao.aln);
        align_wall += wall_ms() - a0;
        if (g_trace & 2) fprintf(stderr, "[necat] cns pass: %lu alignments, %.2f ms\n", (unsigned long)m, wall_ms() - a0);
        return opt->rescue_long_indels ? rescue_pass(c, m, res) : NECAT_OK;
    };
    cns::Knobs kn; kn.spec_estimate_extra = g_cns_spec_extra; kn.spec_cover = g_cns_spec_cover;
    cns::Stats st;
    const double w_run = wall_ms();
    if (!ctx->cns_scratch) ctx->cns_scratch = new cns::Scratch();
    const int rc = cns::run(ts, *opt, kn, fn, &st, (cns::Scratch*)ctx->cns_scratch);
    if (g_trace & 2) fprintf(stderr, "[necat] cns host: setup %.2f ms, init %.2f, select %.2f, gather %.2f, replay %.2f ms\n", w_run - w0, st.init_ms, st.select_ms,
                             st.gather_ms, st.replay_ms);
    auto drop = [&]() { for (u8* b : blocks) necat_free(b); };
    {   // the last columns may still be on their way
        const hipError_t e = ctx->stream_copy ? hipStreamSynchronize(ctx->stream_copy) : hipSuccess;
        ctx->copy_pending = false;
        if (e != hipSuccess && !rc) { drop(); return set_err(ctx, NECAT_ERR_DEVICE, "column copy failed: %s", hipGetErrorString(e)); }
    }
    if (rc) { drop(); return rc; }
    necat_cns_result* r = (necat_cns_result*)calloc(1, sizeof(necat_cns_result));
    uint64_t n_ov = 0, n_rg = 0;
    for (auto& T : ts) { n_ov += T.overlaps.size(); n_rg += T.ranges.size() / 2; }
    if (r) {
        r->templates = (necat_cns_template*)calloc(std::max<uint64_t>(1, n_templates), sizeof(necat_cns_template));
        r->overlaps = (necat_cns_overlap*)malloc(std::max<uint64_t>(1, n_ov) * sizeof(necat_cns_overlap));
        r->ranges = (int32_t*)malloc(std::max<uint64_t>(1, n_rg) * 8);
        r->ops = (uint8_t**)malloc(std::max<size_t>(1, blocks.size()) * sizeof(uint8_t*));
    }
    if (!r || !r->templates || !r->overlaps || !r->ranges || !r->ops) {
        drop();
        if (r) { free(r->templates); free(r->overlaps); free(r->ranges); free(r->ops); free(r); }
        return set_err(ctx, NECAT_ERR_MEMORY, "host malloc failed");
    }
    {
        uint64_t ov = 0, rg = 0;
        for (uint64_t t = 0; t < n_templates; ++t) {
            necat_cns_template& o = r->templates[t];
            o.ovlp_begin = ov; o.range_begin = rg;
            ov += ts[t].overlaps.size(); rg += ts[t].ranges.size() / 2;
            o.ovlp_end = ov; o.range_end = rg;
        }
    }
    cns::parallel_for(n_templates, [&](size_t t) {
        const cns::Template& T = ts[t];
        necat_cns_template& o = r->templates[t];
        o.examined = T.examined ? 1 : 0; o.num_can = T.num_can; o.num_ovlps = T.num_ovlps; o.ident_cutoff = T.ident_cutoff;
        if (!T.overlaps.empty()) memcpy(r->overlaps + o.ovlp_begin, T.overlaps.data(), T.overlaps.size() * sizeof(necat_cns_overlap));
        if (!T.ranges.empty()) memcpy(r->ranges + 2 * o.range_begin, T.ranges.data(), T.ranges.size() * 4);
    });
    r->n_templates = n_templates; r->n_overlaps = n_ov; r->n_ranges = n_rg;
    r->n_ops_blocks = (uint32_t)blocks.size();
    for (size_t b = 0; b < blocks.size(); ++b) r->ops[b] = blocks[b];
    r->n_aligned = st.n_aligned; r->n_used = st.n_used; r->n_rounds = st.n_rounds;
    r->device_ms = device_ms; r->host_ms = wall_ms() - w0 - align_wall - rescue_ms;
    r->n_rescue_tried = n_rescue_tried; r->n_rescued = n_rescued; r->rescue_ms = rescue_ms;
    if (g_trace & 2) fprintf(stderr, "[necat] cns total %.2f ms: passes %.2f (device events %.2f), host %.2f\n", wall_ms() - w0, align_wall, device_ms, r->host_ms);
    ctx->tm.extend_ms = device_ms;
    *out = r;
    return NECAT_OK;
}

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
            if (full) hipLaunchKernelGGL((k_ext_frag<kWordsA, kTWordsA>), dim3(grid_for((u64)g * 64 * (kWordsA + kTWordsA), 256)), dim3(256), 0, s, dv, dv, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, d_frag, RoundCtl());
            else hipLaunchKernelGGL((k_ext_frag<kWordsB, kTWordsB>), dim3(grid_for((u64)g * 64 * (kWordsB + kTWordsB), 256)), dim3(256), 0, s, dv, dv, (const BlockItem*)d_items, m, (const u32*)nullptr, 0u, d_frag, RoundCtl());
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
                                       d_res, d_stats, epoch, 0u, g * 64);
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
                                       d_res, d_stats, epoch, 0u, g * 64);
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
}

}  // extern "C"
