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

// Two builds of these sources.  libnecat_hip.so - the product - launches the kernels of the default paths only; the kernel families those paths replaced and the tests keep
// as independent implementations of the same results (k_myers_coop / k_myers with the band-record k_traceback and k_walk_wave, k_myers_a16, k_rcwalk2, k_rcwalk4, k_myers_ck
// without carries, the lane-per-strand seed collection, the global-atomic index passes, k_asm_align, and the batch Edlib_align hook that drives them block by block) are
// compiled with -DNECAT_BUILD_CROSSCHECK into libnecat_hip_xcheck.so, which the tests' alternative-path cases load (necat_amd/capi.py).  A knob that selects such a path
// in the product build fails the call with NECAT_ERR_ARG and says so - it never falls back to another path.
#ifdef NECAT_BUILD_CROSSCHECK
#define NECAT_XCHECK 1
#else
#define NECAT_XCHECK 0
#endif

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
// when the runtime initialises (the first HIP call of the process): the command-line programs (necat_cli_env, pm_job.h) and bench.py set it before their first HIP
// call; the LIBRARY does not touch its host's environment (until round 6 a load-time constructor did: a setenv behind the back of a multi-threaded host, and without
// effect where the runtime was already up).  A context that finds fewer than 8 says so once under NECAT_TRACE.

namespace necat { thread_local const Knobs* tl_knobs = nullptr; }      // knobs.h: set by KnobScope in every entry point that takes a context
using namespace necat;

#define NECAT_RETIRED(ctx, what) return necat::set_err(ctx, NECAT_ERR_ARG, "%s: a cross-check path this library is built without (it is in libnecat_hip_xcheck.so, -DNECAT_BUILD_CROSSCHECK)", what)

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
#if NECAT_XCHECK
    else hipLaunchKernelGGL((k_rcwalk2<NW, TW, COLS, MAXOPS>), dim3((nitems + 15) / 16), dim3(64), 0, s, a...);
#endif
    // (NECAT_RC_WW=0 in the product build: necat_ctx_create refuses it - read_knobs)
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
    K.ckr_fast = (u32)num("NECAT_CKR_FAST", 1);
    K.ext_lanes = (u32)std::min<unsigned long long>(kMaxExtLanes, std::max<unsigned long long>(1, num("NECAT_EXT_LANES", 2)));
    K.rc_prio = (u32)num("NECAT_RC_PRIO", 1);
    K.rc_pipe = (u32)std::min<unsigned long long>(8, std::max<unsigned long long>(1, num("NECAT_RC_PIPE", 1))); K.rc_pipe_min = (u32)num("NECAT_RC_PIPE_MIN", 49152);
    K.rc_merge = (u32)num("NECAT_RC_MERGE", 1);
    K.frag_fuse = (u32)num("NECAT_FRAG_FUSE", 1);
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

// why the last necat_ctx_create of this thread failed (there is no context to hold that text): necat_last_error(NULL)
static thread_local char g_create_err[256] = "no context";
#define NECAT_CREATE_FAIL(code, ...) do { snprintf(g_create_err, sizeof g_create_err, __VA_ARGS__); return (code); } while (0)

int necat_ctx_create(int device_id, necat_ctx** out)
{
    if (!out) return NECAT_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); NECAT_CREATE_FAIL(NECAT_ERR_DEVICE, "no usable HIP device (there is no CPU fallback)"); }
    if (device_id < 0 || device_id >= ndev) NECAT_CREATE_FAIL(NECAT_ERR_ARG, "device %d of %d", device_id, ndev);
    if (hipSetDevice(device_id) != hipSuccess) NECAT_CREATE_FAIL(NECAT_ERR_DEVICE, "hipSetDevice(%d) failed", device_id);
    necat_ctx* ctx = new necat_ctx();
    ctx->device = device_id;
    read_knobs(ctx->knobs);
#if !NECAT_XCHECK
    if (ctx->knobs.rc_ww == 0) {      // (the one retired path chosen inside a launcher that cannot fail: refused here)
        fprintf(stderr, "[necat] NECAT_RC_WW=0 selects k_rcwalk2, a cross-check kernel this library is built without (libnecat_hip_xcheck.so has it)\n");
        delete ctx; NECAT_CREATE_FAIL(NECAT_ERR_ARG, "NECAT_RC_WW=0 selects k_rcwalk2, a cross-check kernel this library is built without (libnecat_hip_xcheck.so has it)");
    }
#endif
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
    if (hipHostMalloc(&ctx->round_ring, kMaxExtLanes * kRoundRing * sizeof(RoundPub), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&ctx->round_ring_dev, ctx->round_ring, 0) != hipSuccess) { delete ctx; return NECAT_ERR_DEVICE; }
    memset(ctx->round_ring, 0, kMaxExtLanes * kRoundRing * sizeof(RoundPub));      // (a stretch per lane: ExtLane1)
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
    for (auto& lx : ctx->lanex) {
        for (auto& b : lx.buf) if (b.p) (void)hipFree(b.p);
        for (int i = 0; i < kNumEvents; ++i) if (lx.ev[i]) (void)hipEventDestroy(lx.ev[i]);          // (every event that exists, also those of a creation that failed half-way)
        for (hipStream_t st : lx.st) if (st) (void)hipStreamDestroy(st);
    }
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
    for (auto& lx : ctx->lanex) for (auto& b : lx.buf) if (b.p) { (void)hipFree(b.p); b = DevBuf(); }
    ctx->seed_ht_ptr = nullptr; ctx->seed_ht_clean = 0; ctx->seed_ht_cap = 0;
}

const char* necat_last_error(const necat_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

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

int necat_get_timings_sized(const necat_ctx* ctx, void* t, size_t bytes)
{
    if (!ctx || !t) return NECAT_ERR_ARG;
    memcpy(t, &ctx->tm, bytes < sizeof(necat_timings) ? bytes : sizeof(necat_timings));
    return NECAT_OK;
}

int necat_abi_version(void) { return NECAT_ABI_VERSION; }

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

// ---- the stages, one file each: same translation unit, included here in dependency order
#include "stage_volumes.inl"
#include "stage_index.inl"
#include "stage_seed.inl"
#include "stage_extend.inl"
#include "stage_asm_align.inl"
#include "stage_asm_plan.inl"
#include "stage_refmap.inl"
#include "stage_pcan.inl"
#include "stage_multi.inl"
#include "stage_align_batch.inl"
#include "stage_cns.inl"
#include "stage_edlib_batch.inl"

}  // extern "C"
