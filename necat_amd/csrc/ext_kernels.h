// ext_kernels.h - __global__ shells of the extension stage (onc_align for 10^5..10^6 candidates at once).
//
// Every candidate is a small state machine (ext_core.h) that needs one block alignment per round;
// its scheduled block sits in list A (blocks of at most 512 x 512) or list B (bigger last blocks of an
// extension, <= 794 x 794).  A round is three launches per list (the host loop: BatchRun in stage_extend.inl):
//
//   k_ext_frag      gather the two fragments of every block from the 2-bit volumes: query as two
//                   complemented bit-planes per 64 rows, target 2-bit packed; written lane-interleaved
//   k_myers_coop<G> G lanes = one block alignment (anti-diagonal wavefront over the 64-row words, DPP carry,
//                   LDS-staged target bit-planes): SHW pass, then the NW pass that stores the band - or, for
//                   small lists, one storing pass (SINGLE).  The default DP kernel.
//   k_myers<NW>     lane = one block alignment, the reference's banded NW pass line by line (dp_core.h); the
//                   first design, kept as the second implementation the tests compare against
//   k_traceback     walk the stored band back (up > left > diagonal), trim the block tail at the last
//                   run of 8 matches, fold the kept columns into the candidate's running counters, then
//                   plan the candidate's next block and append it to the next round's lists
//
// The NW band is stored as 16-byte records (the walk's decisions, dp_core.h) [column / 8][word][lane][column % 8]: every traceback step is
// one 16-byte load.  Groups of 64 work items share one slab of a band pool; list A (512 cols x 8 words) and list B
// (794 x 13) have pools of their own.
#pragma once
#include "dp_core.h"
#include "ext_core.h"

namespace necat {

constexpr int kColsA = 512, kWordsA = 8, kTWordsA = 16, kOpsA = 1024;
constexpr int kColsB = kMaxFragLen, kWordsB = kMaxWords, kTWordsB = kMaxTWords, kOpsB = 1600;
constexpr int kFragWordsA = 2 * kWordsA + kTWordsA;   // 32 u64 per item
constexpr int kFragWordsB = 2 * kWordsB + kTWordsB;   // 51 u64 per item

// One band record per (column, word, lane): the traceback's decision at each of the word's 64 cells, two bits per
// cell (cell_codes, dp_core.h).  16 bytes, no validity tag: the walk only ever stands on
// cells of the optimal alignment, and the DP kernels store every word such a cell can be in.
// (History: separate P/M, score and band arrays cost four DRAM sectors per traceback step - 5.7 us per step at
// 200 k concurrent blocks; one self-contained 32-byte record with scores and an epoch tag cost one; the bit-rule
// walk halved that again.)
struct __attribute__((aligned(16))) BandRec { u64 A, B; };
static_assert(sizeof(BandRec) == 16, "BandRec must be 16 bytes");
// bytes of one 64-item slab
constexpr size_t kSlabA = (size_t)((kColsA + 7) & ~7) * kWordsA * 64 * sizeof(BandRec);
constexpr size_t kSlabB = (size_t)((kColsB + 7) & ~7) * kWordsB * 64 * sizeof(BandRec);

struct BlockItem {       // one scheduled block alignment
    FragGeom g;
    i32 task;            // owning ExtTask (-1 for the stand-alone batch API)
    i16 qn, tn;
};

struct BlockResult { i32 dist, endc, err; u32 words; };

// The work counters of the extension kernels (word updates, bases, band words, walk blocks / words: necat_timings).  Every wave of every DP /
// walk kernel adds to them, and atomics on ONE cache line are served one after the other, ~ 8 ns each on MI355X whether or not anybody waits
// for the result: 3 per wave of k_myers_ck were 86 k per launch of a big round = 0.65 ms, more than half of that kernel's own 1.2 ms
// (tools/ck_microbench.hip; profiles/NOTES_r04.md 10).  So the counters live in kStatSlots copies on lines of their own, a workgroup adds to
// the copy its index hashes to, and the host sums the copies when it reads them.
constexpr int kStatSlots = 64, kStatStride = 16;                 // copies; u64 per copy (one 128-byte line)
constexpr size_t kStatBytes = (size_t)kStatSlots * kStatStride * 8;
NECAT_D void stat_add(unsigned long long* stats, int idx, unsigned long long v)
{
    atomicAdd(&stats[(size_t)((blockIdx.x * 0x9E3779B1u) >> 26) * kStatStride + idx], v);
}
static_assert(kStatSlots == 64, "stat_add takes the top 6 bits of the hash");
// ext_rcwalk.h: a full block whose walk was done by k_rcwalk4 leaves its statistics here (k_traceback<WALK = 5> takes them instead of
// walking); a block that kernel cannot take (distance too large for its 4-word window) carries kWideFlag in BlockResult::words and
// goes through the DP + walk kernels below in `only wide` launches.
constexpr u32 kWideFlag = 0x80000000u;
struct WalkOut { i32 n, nmat, m, hit, acnt, qcnt, tcnt, mcnt; };

struct ExtLists {
    u32* count;          // [0] = full blocks of list A (front), [1] = nB, [2] = other list-A blocks (back)
    BlockItem* itemsA; BlockItem* itemsB;
    u8* task_ops = nullptr;   // per-task alignment columns (necat_onc_align_batch), nullptr = not kept
    u32 capA = 0;        // capacity of itemsA (its back end is itemsA[capA - 1])
};


// Round bookkeeping done by the first kernel of a round's list-A chain (k_ext_frag): the host never synchronises
// with the device inside the round loop (run_batch) - it learns the sizes of the round's lists from a pinned
// ring the kernel publishes to, and sizes the NEXT round's grids from them (a candidate has one block at a time,
// so a round's lists are never longer than what was alive a round earlier); every kernel reads the exact list
// size from device memory and workgroups beyond it exit.
struct RoundPub { u32 nA, nB; unsigned long long seq; };
struct RoundCtl {
    const u32* count = nullptr;     // (nA, nB) of THIS round's lists, final when the kernel starts
    u32* zero = nullptr;            // (nA, nB) of the list buffer the round after next appends to: reset here
    RoundPub* pub = nullptr;        // host-visible slot of this round
    unsigned long long seq = 0;
    u32* zero_bins = nullptr;       // list B chain: the size-sort counters of the slot, reset for their next use
};

// exact number of work items of a launch: host-known (n_dev == nullptr) or read from the device-side list counter
NECAT_D u32 live_count(u32 n_host, const u32* __restrict__ n_dev) { return n_dev ? *n_dev : n_host; }

// List A is filled from both ends (ext_append_block): the FULL blocks (512 x 512) from the front - itemsA[0 .. nf) -, the shorter
// last blocks from the back - itemsA[cap - 1], itemsA[cap - 2], ... - so that the DP kernels' waves of 8 / 16 consecutive work items
// are all-full almost everywhere (they have a faster path for those).  Work index space of a round: [0, nf) the full blocks,
// [nf, nf16) holes (nf16 = nf rounded up to 16), [nf16, nf16 + np) the others; fragments, band slabs, results and op pools are
// addressed by work index.  cap == 0: a plain list of n items (list B, the batch API).  cnt = the list buffer's counters
// [0] = nf, [1] = nB, [2] = np.
struct ListView { u32 n, nf, nf16, cap; };
NECAT_D ListView list_view(u32 n_host, const u32* __restrict__ cnt, u32 cap)
{
    ListView v; v.cap = cap;
    if (!cap) { v.n = cnt ? *cnt : n_host; v.nf = v.n; v.nf16 = v.n; }
    else { v.nf = cnt[0]; v.nf16 = (v.nf + 15u) & ~15u; v.n = v.nf16 + cnt[2]; }
    return v;
}
NECAT_D bool list_item(const ListView& v, const BlockItem* __restrict__ items, u64 i, BlockItem& it)
{
    if (i < v.nf) { it = items[i]; return true; }
    if (i < v.nf16 || i >= v.n) return false;
    it = items[(u64)v.cap - 1 - (i - v.nf16)];
    return true;
}

// Append the scheduled block of task `ti` to list A (blocks of at most 512 x 512: the full blocks of an
// extension and the last blocks that fit - 8 words, 8 lanes per block) or list B (bigger last blocks, up to
// 794 x 794 - 13 words, 16 lanes per block: 3x the cost, so nothing that fits list A goes here);
// one atomic per wave and list (ballot-aggregated).
// BLOCK: the full block size (512; 2048 in the overlapper of corrected reads).  ONE_LIST: every block goes to list B's arrays, used
// as one plain list (the round loop of necat_asm_align_batch).
template <int BLOCK = kOcaBlockSize, bool ONE_LIST = false>
NECAT_D void ext_append_block(const ExtTask& t, u32 ti, bool go, const ExtLists& L)
{
    const bool isA = !ONE_LIST && go && t.qblk <= BLOCK && t.tblk <= BLOCK;
    const bool isB = go && !isA;
    // list A from both ends: full blocks from the front, the others from the back (ListView)
    const bool isF = isA && t.qblk == BLOCK && t.tblk == BLOCK;
    const bool isP = isA && !isF;
    const int lane = (int)(threadIdx.x & 63);
    const u64 below = (1ULL << lane) - 1ULL;
    const u64 mF = __ballot(isF), mP = __ballot(isP), mB = __ballot(isB);
    u32 baseF = 0, baseP = 0, baseB = 0;
    if (mF) { const int leader = ctz64(mF); if (lane == leader) baseF = atomicAdd(&L.count[0], (u32)popc64(mF)); baseF = __shfl(baseF, leader); }
    if (mP) { const int leader = ctz64(mP); if (lane == leader) baseP = atomicAdd(&L.count[2], (u32)popc64(mP)); baseP = __shfl(baseP, leader); }
    if (mB) { const int leader = ctz64(mB); if (lane == leader) baseB = atomicAdd(&L.count[1], (u32)popc64(mB)); baseB = __shfl(baseB, leader); }
    if (go) {
        BlockItem it;
        it.g = ext_frag_geom(t); it.task = (i32)ti; it.qn = (i16)t.qblk; it.tn = (i16)t.tblk;
        if (isF) L.itemsA[baseF + (u32)popc64(mF & below)] = it;
        else if (isP) L.itemsA[L.capA - 1u - (baseP + (u32)popc64(mP & below))] = it;
        else L.itemsB[baseB + (u32)popc64(mB & below)] = it;
    }
}

// The same for a workgroup of WAVES waves whose every thread calls it: ONE atomic per workgroup and list (the list counters are
// three addresses the whole grid reserves its slots at - 3 k waves x 2 - 3 atomics in a row on them was the finishing kernel's time).
template <int BLOCK, bool ONE_LIST, int WAVES>
NECAT_D void ext_append_block_wg(const ExtTask& t, u32 ti, bool go, const ExtLists& L)
{
    __shared__ u32 wcnt[3][WAVES];
    __shared__ u32 wbase[3];
    const bool isA = !ONE_LIST && go && t.qblk <= BLOCK && t.tblk <= BLOCK;
    const bool isB = go && !isA;
    const bool isF = isA && t.qblk == BLOCK && t.tblk == BLOCK;
    const bool isP = isA && !isF;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const u64 below = (1ULL << lane) - 1ULL;
    const u64 mF = __ballot(isF), mP = __ballot(isP), mB = __ballot(isB);
    if (lane == 0) { wcnt[0][wave] = (u32)popc64(mF); wcnt[1][wave] = (u32)popc64(mP); wcnt[2][wave] = (u32)popc64(mB); }
    __syncthreads();
    if (threadIdx.x < 3) {
        u32 tot = 0;
        for (int w = 0; w < WAVES; ++w) tot += wcnt[threadIdx.x][w];
        wbase[threadIdx.x] = tot ? atomicAdd(&L.count[threadIdx.x == 0 ? 0 : threadIdx.x == 1 ? 2 : 1], tot) : 0u;
    }
    __syncthreads();
    u32 baseF = wbase[0], baseP = wbase[1], baseB = wbase[2];
    for (int w = 0; w < wave; ++w) { baseF += wcnt[0][w]; baseP += wcnt[1][w]; baseB += wcnt[2][w]; }
    if (go) {
        BlockItem it;
        it.g = ext_frag_geom(t); it.task = (i32)ti; it.qn = (i16)t.qblk; it.tn = (i16)t.tblk;
        if (isF) L.itemsA[baseF + (u32)popc64(mF & below)] = it;
        else if (isP) L.itemsA[L.capA - 1u - (baseP + (u32)popc64(mP & below))] = it;
        else L.itemsB[baseB + (u32)popc64(mB & below)] = it;
    }
}

__global__ void __launch_bounds__(256)
k_ext_init(const necat_candidate* __restrict__ cands, u32 n, u32 cand_base, int read_start_id, int ref_start_id,
           const u64* __restrict__ reads_off, const u64* __restrict__ ref_off, ExtTask* __restrict__ tasks, ExtLists L,
           const u64* __restrict__ ops_base, const u32* __restrict__ perm, int window)
{
    // task i of the batch = candidate perm[cand_base + i] (batches ordered by expected chain length) or cand_base + i
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    ExtTask t;
    bool go = false;
    if (i < n) {
        const u32 ci = perm ? perm[cand_base + i] : cand_base + i;
        const necat_candidate c = cands[ci];
        const int lq = c.qid - read_start_id, ls = c.sid - ref_start_id;
        if (window) {      // read-to-reference mapping: the subject is the stretch rm_window gives, not the whole sequence
            i64 from, to, woff;
            rm_window((i64)c.qoff, (i64)c.qsize, (i64)c.soff, (i64)c.ssize, &from, &to, &woff);
            ext_init(t, (i32)ci, c.qdir, (i64)reads_off[lq], (i32)c.qsize, (i64)ref_off[ls] + from, (i32)(to - from), (i32)c.qoff, (i32)woff);
        } else
        ext_init(t, (i32)ci, c.qdir, (i64)reads_off[lq], (i32)c.qsize, (i64)ref_off[ls], (i32)c.ssize, (i32)c.qoff, (i32)c.soff);
        if (ops_base) t.ops_base = ops_base[i];
        go = ext_plan(t);          // first block (or an immediately finished candidate)
        tasks[i] = t;
    }
    ext_append_block_wg<kOcaBlockSize, false, 4>(t, i, go, L);      // (256 threads: one reservation per workgroup and list)
}

// ---- batch order: with several batches the candidates are dealt out by expected chain length (what is left of the two reads beyond
// the anchor, in 480-base blocks), longest first.  MODE 0: candidates per length bin.  MODE 1: perm[cursor[bin]++] = candidate
// (one reservation per (block, bin); the order inside a bin is free - a candidate's records do not depend on its batch).
constexpr int kLenBins = 128;
NECAT_D u32 chain_len_bin(const necat_candidate& c)
{
    const u64 r1 = c.qsize - c.qoff, r2 = c.ssize - c.soff, right = r1 < r2 ? r1 : r2, left = c.qoff < c.soff ? c.qoff : c.soff;
    const u64 b = right / 480 + left / 480;
    return (u32)(kLenBins - 1) - (u32)(b < (u64)(kLenBins - 1) ? b : (u64)(kLenBins - 1));     // longest first
}
template <int MODE>
__global__ void __launch_bounds__(256)
k_len_order(const necat_candidate* __restrict__ cands, u32 n, u32* __restrict__ cursor, u32* __restrict__ perm)
{
    __shared__ u32 cnt[kLenBins], base[kLenBins];
    if (threadIdx.x < kLenBins) cnt[threadIdx.x] = 0;
    __syncthreads();
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    u32 bin = 0, rk = 0;
    if (i < n) { bin = chain_len_bin(cands[i]); rk = atomicAdd(&cnt[bin], 1u); }
    __syncthreads();
    if (threadIdx.x < kLenBins) { const u32 c = cnt[threadIdx.x]; base[threadIdx.x] = c ? atomicAdd(&cursor[threadIdx.x], c) : 0u; }
    if (MODE == 1) {
        __syncthreads();
        if (i < n) perm[base[bin] + rk] = i;
    }
}

// ---- list B ordering: blocks of list B have any size up to 794 x 794; lanes (k_traceback, k_myers) or
// lane groups (k_myers_coop) of one wave finish together only if their blocks are alike, so the list is
// counting-sorted by target length (longest first) before the round's kernels run.
constexpr int kSortBins = kMaxFragLen + 1;

#if NECAT_XCHECK          // the size sort of list B exists for the band-record path only: cross-check build only (necat_hip.hip)
__global__ void __launch_bounds__(256)
k_items_hist(const BlockItem* __restrict__ items, u32 n, u32* __restrict__ bins)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&bins[kMaxFragLen - items[i].tn], 1u);     // bin 0 = longest
}

// one wave, no barrier: lane l owns kBinsPerLane consecutive bins.  (As a 1024-thread workgroup with a barrier per scan step this
// 5 us kernel took 0.6 - 1.5 ms whenever the DP kernel of list A filled the chip: its 16 waves queued behind that kernel's at
// every barrier.)
constexpr int kBinsPerLane = (kSortBins + 63) / 64;
__global__ void __launch_bounds__(64)
k_items_scan(u32* __restrict__ bins)          // exclusive scan of kSortBins counters, in place
{
    const int lane = threadIdx.x;
    u32 v[kBinsPerLane], sum = 0;
#pragma unroll
    for (int q = 0; q < kBinsPerLane; ++q) { const int i = lane * kBinsPerLane + q; v[q] = i < kSortBins ? bins[i] : 0u; sum += v[q]; }
    u32 incl = sum;
    for (int o = 1; o < 64; o <<= 1) { const u32 x = __shfl_up(incl, o); if (lane >= o) incl += x; }
    u32 run = incl - sum;
#pragma unroll
    for (int q = 0; q < kBinsPerLane; ++q) { const int i = lane * kBinsPerLane + q; if (i < kSortBins) bins[i] = run; run += v[q]; }
}

__global__ void __launch_bounds__(256)
k_items_scatter(const BlockItem* __restrict__ items, u32 n, u32* __restrict__ bins, BlockItem* __restrict__ out)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const BlockItem it = items[i]; out[atomicAdd(&bins[kMaxFragLen - it.tn], 1u)] = it; }
}
#endif

constexpr int kFragSplit = 4;          // threads per item of k_ext_frag (each takes every kFragSplit-th channel)
// frag layout per 64-item group g: word w of lane l at frag[(g * FW + w) * 64 + l];
// words [0,NW) = ~lo planes, [NW,2NW) = ~hi planes, [2NW, 2NW+TW) = target 2-bit words
template <int NW, int TW>
__global__ void __launch_bounds__(256)
k_ext_frag(DevVolume reads, DevVolume ref, const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, u64* __restrict__ frag,
           RoundCtl ctl)
{
    constexpr int FW = 2 * NW + TW, CH = NW + TW;
    const ListView lv = list_view(n_host, n_dev, capA);
    if (blockIdx.x == 0) {
        if (ctl.zero_bins) for (int i = threadIdx.x; i < 1024; i += blockDim.x) ctl.zero_bins[i] = 0u;
        if (threadIdx.x == 0 && ctl.pub) {
            const u32 a = ctl.count[0] + ctl.count[2], b = ctl.count[1];      // list A: full blocks + the others
            ctl.zero[0] = 0u; ctl.zero[1] = 0u; ctl.zero[2] = 0u;
            volatile RoundPub* p = ctl.pub;
            p->nA = a; p->nB = b;
            __threadfence_system();
            p->seq = ctl.seq;
            __threadfence_system();
        }
    }
    // Round 6: a thread takes ITS item's channels kFragSplit apart (6 of the 24 of a list-A item) instead of one channel - the 40-byte item is read by 4 threads, not by
    // 24, and a thread has 6 independent volume loads in flight where it had one: 54 -> ~ 30 us per 222 k-block round (the kernel is a chain of two dependent loads per
    // thread - item, then bases - and what it lacked was loads in flight per wave, not waves).  The grid covers items x kFragSplit threads.
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 grp = gid / (64 * kFragSplit);
    const u32 r = (u32)(gid % (64 * kFragSplit));
    const int part = (int)(r >> 6), lane = (int)(r & 63);
    const u64 item = grp * 64 + lane;
    BlockItem it;
    if (!list_item(lv, items, item, it)) return;
    u64* dst = frag + grp * FW * 64 + lane;
#pragma unroll
    for (int ch = part; ch < CH; ch += kFragSplit) {
        if (ch < NW) {
            if (ch * 64 < it.qn) {
                u64 lo, hi;
                load64_planes(reads.bases, it.g.q_base, it.g.q_dir, it.g.q_comp, ch * 64, &lo, &hi);
                dst[(u64)ch * 64] = ~lo; dst[(u64)(NW + ch) * 64] = ~hi;
            }
        } else {
            const int tw = ch - NW;
            if (tw * 32 < it.tn)
                dst[(u64)(2 * NW + tw) * 64] = load32_dir(ref.bases, it.g.t_base + (i64)it.g.t_dir * (tw * 32), it.g.t_dir, it.g.t_comp);
        }
    }
}

template <int NW>
struct TgtReader {
    const u64* w; u64 cur;
    NECAT_D int code(int c)
    {
        if ((c & 31) == 0) cur = w[(u64)(c >> 5) * 64];
        return (int)((cur >> ((c & 31) * 2)) & 3);
    }
};

// Position of record (column c, word b) of a lane inside its slab, in 16-byte units.  Eight consecutive
// columns of one (word, lane) share a 128-byte line: the traceback walks column by column at a (mostly) fixed
// word, so a line it pulls from HBM serves eight steps, and the cooperative DP kernel - one lane per word, one
// column per step - fills that line with eight consecutive 16-byte stores of the same lane.
// (With the records of the 64 lanes interleaved per column every traceback step was its own DRAM access.)
// (32-bit: a slab is < 16 MiB, and a wave-uniform slab base + 32-bit lane offset is the cheap addressing mode)
template <int NW>
NECAT_D u32 rec_pos(int c, int b, int lane)
{
    return ((((u32)c >> 3) * (u32)NW + (u32)b) * 64u + (u32)lane) * 8u + ((u32)c & 7u);
}

template <int NW>
struct MatWriter {
    ulonglong2* rec;   // slab base (16-byte units)
    int lane;
    int dbg;
    NECAT_D bool skip_nw() const { return dbg == 2; }
    NECAT_D void store(int c, int b, u64 Pv, u64 Ph)
    {
        if (dbg == 1) return;
        rec[rec_pos<NW>(c, b, lane)] = make_ulonglong2(Pv, Ph);
    }
};

// Band reader of the traceback.  The walk moves one column at a time at a (mostly) fixed word b - a
// chain of dependent loads.  The reader issues the load of column c-1 while column c is consumed, so
// the walk pays max(compute, latency) per column instead of their sum.
template <int NW>
struct MatReader {
    const ulonglong2* base;    // slab base
    int lane;
    int nc, nb;                // coordinates of the prefetched record
    ulonglong2 nv;
    bool no_prefetch = false;  // A/B switch (NECAT_WALK=2)
    bool nt = false;           // A/B switch (NECAT_WALK=3 / 4): non-temporal loads (the record lines are read once, by one lane)
    NECAT_D void init() { nc = -100; nb = -100; }
    NECAT_D ulonglong2 ld(u32 pos) const
    {
        if (!nt) return base[pos];
        typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
        const v2 x = __builtin_nontemporal_load(reinterpret_cast<const v2*>(base) + pos);
        return make_ulonglong2(x.x, x.y);
    }
    NECAT_D void rec(int c, int b, u64& Pv, u64& Ph)
    {
        if (no_prefetch) { const ulonglong2 v = ld(rec_pos<NW>(c, b, lane)); Pv = v.x; Ph = v.y; return; }
        ulonglong2 v = nv;
        if (!(c == nc && b == nb)) v = ld(rec_pos<NW>(c, b, lane));
        Pv = v.x; Ph = v.y;
        nc = c - 1; nb = b;
        if (nc >= 0) nv = ld(rec_pos<NW>(nc, nb, lane));
    }
};

NECAT_D ulonglong2* slab_records(char* slab) { return reinterpret_cast<ulonglong2*>(slab); }

// The hot kernel.  One wave = 64 block alignments in lock-step.  No LDS: the whole column state is
// register resident (dp_core.h) and the only memory traffic is the coalesced band store.
template <int NW, int TW, int COLS, bool FULL>
__global__ void __launch_bounds__(64)
k_myers(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, char* __restrict__ slabs, size_t slab_bytes,
        double error, BlockResult* __restrict__ results, unsigned long long* __restrict__ stats, u32 epoch, u32 item_base)
{
    constexpr int FW = 2 * NW + TW;
    const ListView lv = list_view(n_host, n_dev, capA);
    const u32 n = lv.n;
    const u32 grp = blockIdx.x + (item_base >> 6);      // item_base is a multiple of 64; n = end of this launch's range
    const int lane = threadIdx.x;
    const u64 item = (u64)grp * 64 + lane;
    BlockItem it0;
    if (!list_item(lv, items, item, it0)) return;
    const int qn = FULL ? kOcaBlockSize : it0.qn;
    const int tn = FULL ? kOcaBlockSize : it0.tn;
    const u64* fr = frag + (u64)grp * FW * 64 + lane;
    MyersRegs<NW> R;
    const int nblk = (qn + 63) >> 6;
#pragma unroll
    for (int b = 0; b < NW; ++b) {
        const bool have = FULL || b < nblk;
        R.nlo[b] = have ? fr[(u64)b * 64] : 0ULL;
        R.nhi[b] = have ? fr[(u64)(NW + b) * 64] : 0ULL;
    }
    TgtReader<NW> tg; tg.w = fr + (u64)2 * NW * 64; tg.cur = 0;
    MatWriter<NW> mw;
    mw.rec = slab_records(slabs + (size_t)grp * slab_bytes); mw.lane = lane; mw.dbg = (int)(epoch >> 28);
    const MyersResult r = myers_block<NW, FULL>(R, qn, tn, error, tg, mw);
    BlockResult br; br.dist = r.dist; br.endc = r.endc; br.err = r.err; br.words = r.words;
    results[item] = br;
    // work counters for the roofline report: one atomic per wave
    u32 w = r.words, bases = (u32)(qn + tn);
    if (__popcll(__ballot(1)) == 64) {       // full wave: every lane is alive, shuffles are safe
        for (int o = 32; o > 0; o >>= 1) { w += __shfl_down(w, o); bases += __shfl_down(bases, o); }
        if (lane == 0) { stat_add(stats, 0, (unsigned long long)w); stat_add(stats, 1, (unsigned long long)bases); }
    } else { stat_add(stats, 0, (unsigned long long)w); stat_add(stats, 1, (unsigned long long)bases); }
}

// Cooperative variant for rounds with few blocks (the latency-bound tail: a candidate's blocks form a
// dependent chain, so once only the longest overlaps are left the chip is empty and only the latency
// of ONE block alignment matters).  G lanes share a block, lane b owns 64-row word b, and column c of
// word b is computed at step c + b (anti-diagonal wavefront); the carry between vertically adjacent
// words travels by DPP row_shr:1, no LDS.  The band heuristics are dropped: every word of every
// column is computed and stored ("band" = all words).  Distance, end column and traceback are
// unchanged by that - banding only prunes cells that cannot lie on an alignment of cost <= k
// (Ukkonen), every cell the traceback visits or compares against is exact in both, and any cell
// whose value differs is > k - which tests/test_gpu_parity.py::test_coop_equals_banded checks.
NECAT_D int dpp_from_lane_below(int v)   // lane i receives v of lane i-1 (within a row of 16); row lane 0 gets 1
{
    return __builtin_amdgcn_update_dpp(1, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
}
// the same for lane groups wider than a DPP row (the 32 / 44 words of a 2048-bp block): wave_shr:1, lane i receives v of lane i - 1
// across the whole wave (GFX9 DPP control 0x138); wave lane 0 gets 1
template <int G>
NECAT_D int lane_below(int v)
{
    if (G <= 16) return dpp_from_lane_below(v);
    return __builtin_amdgcn_update_dpp(1, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}


// ---------------------------------------------------------------------------------------------------------------
// Fast path of the list-A DP kernel: every block of the wave is a FULL block (512 x 512, 8 words, no pad rows).
// Same recurrence, same results; what changes is how it is issued.  Measured on MI355X (tools/valu_microbench2.hip,
// profiles/r02_valu_microbench2.txt): 32-bit add / and / or / xor / v_bitop3 issue in 2 SIMD cycles per wave64, but
// shifts, v_alignbit, v_bfe, DPP moves, compares, v_cndmask, carry adds and every 64-bit op take 3.5 - and a 64-bit
// v_lshl_add_u64 costs one such slot where the add_co / addc pair costs two.  So here:
//   * (Eq & Pv) + Pv is ONE 64-bit add;
//   * the horizontal carries travel as the raw high words of Ph / Mh (two DPP moves) and enter the next word's
//     shift-by-one through the funnel shift itself (v_alignbit takes bit 31 of the neighbour's word): no hout
//     arithmetic, no +1 / -1 decoding, no select for the top word - the lane of word 7 publishes the constant
//     boundary carry (+1) for the top word of the next block in its DPP row;
//   * the target stream of lane b is skewed by b columns when a 32-column window is loaded, so every lane reads bit
//     (step & 31) of its window: the reload is wave-uniform (a scalar branch, no exec-mask games), and the step loop has
//     a scalar trip count;
//   * no per-step predicate: all 64 lanes are inside their blocks for steps 7 .. 511; the 7 fill / 7 drain steps run
//     the same body under a lane mask;
//   * the SHW result (smallest bottom-row value, FIRST column attaining it) is one running minimum of
//     (score << 10 | step) - no compare / branch chain;
//   * in the NW pass lanes past their block's end column keep computing (nothing reads them) and only the store is
//     masked.
// The sanity re-derivation of the distance at the end of the NW pass (err = 2 in the general path) is not repeated here.
struct FastWord {
    u64 Pv, Mv;
    u32 pubP, pubM;      // high words of Ph / Mh of this lane's last step: what the lane below reads (bit 31 = carry +1 / -1)
};

// v_bitop3_b32: any function of three words in one full-rate instruction.  IMM = the function evaluated on a = 0xf0, b = 0xcc, c = 0xaa.
template <unsigned IMM> NECAT_D u32 bop(u32 a, u32 b, u32 c) { return (u32)__builtin_amdgcn_bitop3_b32(a, b, c, IMM); }

template <bool REC>
NECAT_D void fast_advance(FastWord& w, u32 el, u32 eh, u32 cph, u32 cmh, u32 cm, u32& phh_out, u32& mhh_out, u64& A, u64& B, u32* phl_out = nullptr, u32* mhl_out = nullptr)
{
    const u32 pl = (u32)w.Pv, ph = (u32)(w.Pv >> 32), ml = (u32)w.Mv, mh = (u32)(w.Mv >> 32);
    const u32 xvl = el | ml, xvh = eh | mh;                        // Xv = Eq | Mv (before the hin fix-up of Eq, edlib_ex.c:71-106)
    const u32 e2l = el | (cmh >> 31);                              // hin == -1
    const u64 sum = (((u64)(eh & ph) << 32) | (e2l & pl)) + w.Pv;  // ONE 64-bit add (v_lshl_add_u64)
    const u32 sl = (u32)sum, sh = (u32)(sum >> 32);
    const u32 xhl = bop<0xbe>(sl, pl, e2l), xhh = bop<0xbe>(sh, ph, eh);      // Xh = (sum ^ Pv) | Eq
    const u32 phl = bop<0xf1>(ml, xhl, pl), phh = bop<0xf1>(mh, xhh, ph);      // Ph = Mv | ~(Xh | Pv)
    const u32 mhl = pl & xhl, mhh = ph & xhh;                                  // Mh = Pv & Xh
    phh_out = phh; mhh_out = mhh;
    if (phl_out) { *phl_out = phl; *mhl_out = mhl; }               // (the ragged fast path tracks a row that is not the word's last)
    w.pubP = phh | cm; w.pubM = mhh & ~cm;                         // word 7 publishes the top-row boundary (+1) for the next block
    const u32 p2l = __builtin_amdgcn_alignbit(phl, cph, 31), p2h = __builtin_amdgcn_alignbit(phh, phl, 31);   // (Ph << 1) | (hin == +1)
    const u32 m2l = __builtin_amdgcn_alignbit(mhl, cmh, 31), m2h = __builtin_amdgcn_alignbit(mhh, mhl, 31);   // (Mh << 1) | (hin == -1)
    const u32 ol = bop<0xf1>(m2l, xvl, p2l), oh = bop<0xf1>(m2h, xvh, p2h);    // Pv' = Mh | ~(Xv | Ph)
    const u32 nl = p2l & xvl, nh = p2h & xvh;                                  // Mv' = Ph & Xv
    if (REC) {    // the walk's decision per cell (dp_core.h: advance_block_rec): A = Pv' | (Pv & ~Xh), B = ~Pv' & (Mv | ~Xh)
        A = ((u64)bop<0xf4>(oh, ph, xhh) << 32) | bop<0xf4>(ol, pl, xhl);
        B = ((u64)bop<0x0d>(oh, mh, xhh) << 32) | bop<0x0d>(ol, ml, xhl);
    }
    w.Pv = ((u64)oh << 32) | ol; w.Mv = ((u64)nh << 32) | nl;
}

// advance_block / advance_block_rec (dp_core.h) issued like fast_advance - one 64-bit add, 3-input logic through v_bitop3 - for the
// general path (ragged blocks, list B, the single-pass kernel of the small rounds), whose horizontal carry is the int hin / hout
// in {-1, 0, +1} of the reference (edlib_ex.c:71-106).  Same results bit for bit.
template <bool REC>
NECAT_D int advance_dev(u64& Pv, u64& Mv, const u64 Eq, const int hin, u64& A, u64& B)
{
    const u32 pl = (u32)Pv, ph = (u32)(Pv >> 32), ml = (u32)Mv, mh = (u32)(Mv >> 32);
    const u32 el = (u32)Eq, eh = (u32)(Eq >> 32);
    const u32 neg = (u32)hin >> 31;                                // 1 iff hin == -1
    const u32 pos = (u32)(hin + 1) >> 1;                           // 1 iff hin == +1
    const u32 xvl = el | ml, xvh = eh | mh;                        // Xv = Eq | Mv (before the hin fix-up of Eq)
    const u32 e2l = el | neg;
    const u64 sum = (((u64)(eh & ph) << 32) | (e2l & pl)) + Pv;
    const u32 sl = (u32)sum, sh = (u32)(sum >> 32);
    const u32 xhl = bop<0xbe>(sl, pl, e2l), xhh = bop<0xbe>(sh, ph, eh);      // Xh = (sum ^ Pv) | Eq
    const u32 phl = bop<0xf1>(ml, xhl, pl), phh = bop<0xf1>(mh, xhh, ph);      // Ph = Mv | ~(Xh | Pv)
    const u32 mhl = pl & xhl, mhh = ph & xhh;                                  // Mh = Pv & Xh
    const int hout = (int)(phh >> 31) - (int)(mhh >> 31);
    const u32 p2l = (phl << 1) | pos, p2h = __builtin_amdgcn_alignbit(phh, phl, 31);
    const u32 m2l = (mhl << 1) | neg, m2h = __builtin_amdgcn_alignbit(mhh, mhl, 31);
    const u32 ol = bop<0xf1>(m2l, xvl, p2l), oh = bop<0xf1>(m2h, xvh, p2h);    // Pv' = Mh | ~(Xv | Ph)
    const u32 nl = p2l & xvl, nh = p2h & xvh;                                  // Mv' = Ph & Xv
    if (REC) {
        A = ((u64)bop<0xf4>(oh, ph, xhh) << 32) | bop<0xf4>(ol, pl, xhl);      // Pv' | (Pv & ~Xh)
        B = ((u64)bop<0x0d>(oh, mh, xhh) << 32) | bop<0x0d>(ol, ml, xhl);      // ~Pv' & (Mv | ~Xh)
    }
    Pv = ((u64)oh << 32) | ol; Mv = ((u64)nh << 32) | nl;
    return hout;
}

NECAT_D u32 dpp_row_shr1(u32 v, u32 keep)     // lane i receives v of lane i - 1 (within a row of 16); row lane 0 keeps `keep`
{
    return (u32)__builtin_amdgcn_update_dpp((int)keep, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
}

// NW pass of 8 full blocks with 8 lanes per block (lane = 8 sub + b): target[0 .. end0] with k = best, the band stored through the
// reference's per-word band tests.  best / end0 / go are the block's (the same on its 8 lanes).
template <int NW, int TW>
NECAT_D void fast_nw8(const int lane, const u64* __restrict__ tw, const u64 nlo, const u64 nhi, ulonglong2* __restrict__ rec, const int il,
                      const int best, const int end0, const bool go, unsigned long long* __restrict__ stats, const bool no_store)
{
    constexpr int G = 8, N = kOcaBlockSize;
    const int b = lane & (G - 1);
    const u32 cm = b == G - 1 ? 0x80000000u : 0u;
    const u32 nlo_l = (u32)nlo, nlo_h = (u32)(nlo >> 32), nhi_l = (u32)nhi, nhi_h = (u32)(nhi >> 32);
    const u32 sk = (u32)(32 - b) & 31u;
    u32 tlo = 0, thi = 0, plo = 0, phi = 0;
    auto reload = [&](int w) {
        const u64 x = w < TW ? tw[w] : 0ULL;
        const u32 xl = (u32)x, xh = (u32)(x >> 32);
        tlo = b ? __builtin_amdgcn_alignbit(xl, plo, sk) : xl;
        thi = b ? __builtin_amdgcn_alignbit(xh, phi, sk) : xh;
        plo = xl; phi = xh;
    };
    auto eq_of = [&](int j, u32& el, u32& eh) {
        const u32 ma = (u32)__builtin_amdgcn_sbfe((int)tlo, (u32)j, 1u), mb = (u32)__builtin_amdgcn_sbfe((int)thi, (u32)j, 1u);
        el = bop<0x60>(nlo_l ^ ma, nhi_l, mb); eh = bop<0x60>(nlo_h ^ ma, nhi_h, mb);
    };
    const int tn2 = end0 + 1;
    FastWord w;
    u32 cph, cmh;
    // ------------------------------------------------------------------ NW on target[0 .. end0] with k = best: store the band
    int steps = go ? tn2 + G - 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(steps, o); steps = x > steps ? x : steps; }
    steps = __builtin_amdgcn_readfirstlane(steps);
    w.Pv = ~0ULL; w.Mv = 0ULL; w.pubP = 0x80000000u; w.pubM = 0u;
    int Sn = (b + 1) * 64;
    // store filter (the reference's per-word band tests with k = best, edlib_ex.c:311-325, as in the general path), in terms of
    // the step s = c + b:   drop iff S >= K1  ||  S - s > K2  ||  S + s > K3;   nothing is stored from column tn2 on
    const int rb = (b + 1) * 64 - 1;
    const int K1 = best + 64, K2 = best + 127 + N - tn2 - rb - b, K3 = rb + best + tn2 - N + b;
    const int s_end = go ? tn2 + b : 0;            // first step past the block's last column
    // byte offset of record (c, b) of this lane's block in the slab (rec_pos): + 16 per column, + 4089 * 16 when c & 7 wraps -
    // which, c being s - b, happens at a fixed position of every group of 8 steps: one precomputed increment per position
    char* const rbase = reinterpret_cast<char*>(rec);
    auto rec_off = [&](int c) -> u32 { return (((((u32)c >> 3) * (u32)NW + (u32)b) * 64u + (u32)il) * 8u + ((u32)c & 7u)) * 16u; };
    u32 inc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) inc[q] = (((b + 7) & 7) == q) ? (u32)(NW * 64 * 8 - 7) * 16u : 16u;    // from step s (s & 7 == q) to s + 1
    cph = 0x80000000u; cmh = 0u;
    u32 kept = 0;                                  // band words stored by this lane (work counter)
    auto nw_step = [&](int s, int j, u32 off) {
        u32 phh, mhh, el, eh;
        u64 rA, rB;
        eq_of(j, el, eh);
        fast_advance<true>(w, el, eh, cph, cmh, cm, phh, mhh, rA, rB);
        Sn += (int)(phh >> 31) - (int)(mhh >> 31);
        const bool keep = s < s_end && Sn < K1 && Sn - s <= K2 && Sn + s <= K3;
        if (keep && !no_store) { *reinterpret_cast<ulonglong2*>(rbase + off) = make_ulonglong2(rA, rB); ++kept; }
    };
    // the first 8 steps: lane b enters at step b
    reload(0);
    for (int s = 0; s < 8 && s < steps; ++s) {
        cph = dpp_row_shr1(w.pubP, cph); cmh = dpp_row_shr1(w.pubM, cmh);
        if (s >= b) nw_step(s, s, rec_off(s - b));
    }
    // then every lane is inside its block (or past its end: computing on, storing nothing): groups of 8 steps
    u32 off = rec_off(8 - b);
    for (int s0 = 8; s0 < steps; s0 += 8) {
        if ((s0 & 31) == 0) reload(s0 >> 5);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            cph = dpp_row_shr1(w.pubP, cph); cmh = dpp_row_shr1(w.pubM, cmh);
            nw_step(s0 + q, (s0 & 31) + q, off);
            off += inc[q];
        }
    }
    for (int o = 32; o > 0; o >>= 1) kept += (u32)__shfl_xor((int)kept, o);
    if (lane == 0 && kept) stat_add(stats, 2, (unsigned long long)kept);
}

}  // namespace necat
#include "ext_fast16.h"      // fast_shw8, myers_fast16 (on top of FastWord / fast_advance / fast_nw8 above)
namespace necat {

// all 8 blocks of the wave: 512 x 512.  tw = the wave's staged target planes (LDS), nlo / nhi = this lane's query word.
template <int NW, int TW>
NECAT_D void myers_fast_full(const int lane, const u64* __restrict__ tw, const u64 nlo, const u64 nhi, ulonglong2* __restrict__ rec, const int il,
                             const double error, const bool valid_item, BlockResult* __restrict__ result, unsigned long long* __restrict__ stats,
                             const bool no_store)
{
    constexpr int G = 8, N = kOcaBlockSize;
    const int b = lane & (G - 1);
    // ---- SHW: best prefix distance, first best end column (word 7 holds the bottom row; column = step - 7)
    const u32 key = fast_shw8<TW>(b, tw, nlo, nhi);
    const int owner = (lane & ~(G - 1)) | (G - 1);
    const u32 bkey = (u32)__shfl((int)key, owner);
    int best = (int)(bkey >> 10), end0 = (int)(bkey & 1023u) - (G - 1);
    const int k0 = (int)((double)N * error * 1.1);                       // edlib_ex.c:751
    if (best > k0 || !valid_item) best = -1;
    const int tn2 = end0 + 1;
    int err = 0;
    if (best >= 0) { int ad = tn2 - N; if (ad < 0) ad = -ad; if (best < ad) err = 1; }
    const bool go = best >= 0 && !err;
    fast_nw8<NW, TW>(lane, tw, nlo, nhi, rec, il, best, end0, go, stats, no_store);
    if (b == G - 1 && valid_item) {
        BlockResult br; br.dist = err ? -1 : best; br.endc = end0; br.err = err;
        br.words = (u32)(NW * (N + (go ? tn2 : 0)));
        *result = br;
        stat_add(stats, 0, (unsigned long long)br.words); stat_add(stats, 1, (unsigned long long)(2 * N));
    }
}

// SINGLE (small lists, where the round is as long as ONE block alignment): the NW pass recomputes exactly what the
// SHW pass computed for the columns up to the end column (same recurrence, same boundary), its only purpose being
// to know the distance for the store filter - so when store traffic is irrelevant the SHW pass stores every word
// itself and the NW pass is skipped: half the latency.
// The body of one wave: BPW = 64 / G consecutive work items starting at `wave_first`; t_lds = the wave's [BPW][TW] staging area (LDS).
// Every wave of the workgroup must call it (it synchronises the workgroup once).
template <int NW, int TW, int COLS, int G, bool SINGLE>
NECAT_D void myers_coop_wave(const ListView& lv, const BlockItem* __restrict__ items, const u64* __restrict__ frag, char* __restrict__ slabs, size_t slab_bytes,
                             double error, BlockResult* __restrict__ results, unsigned long long* __restrict__ stats, u32 epoch, const u64 wave_first,
                             u64 (*t_lds)[TW], const int lane)
{
    constexpr int FW = 2 * NW + TW, BPW = 64 / G;
    const u32 n = lv.n;
    const int sub = lane / G, b = lane % G;
    const bool filter = ((epoch >> 30) & 1u) == 0;      // bit 30 of the epoch argument switches the store filter off (A/B tests)
    // bit 25: only the blocks k_rcwalk4 left out (kWideFlag) - their band for the old walk; results stay as they are.  bit 24 (the round
    // runs k_myers_ck + k_rcwalk4 on the full blocks, ext_rcwalk.h): that mode for the full-block part of the work index space
    // [0, nf16), the normal one for the rest
    // bit 26: the ragged part only - [nf16, n)
    if (((epoch >> 26) & 1u) && wave_first < (u64)lv.nf16) { __syncthreads(); return; }
    const bool only_wide = ((epoch >> 25) & 1u) != 0 || (((epoch >> 24) & 1u) != 0 && wave_first < (u64)lv.nf16);
    const bool fast_ok = ((epoch >> 29) & 1u) == 0 && !only_wide;     // bit 29: never take the full-block fast path (A/B measurements)
    const bool fast_nostore = ((epoch >> 28) & 1u) != 0; // bit 28: fast path without band stores (profiling only)
    epoch &= 0x07ffffffu;
    const u64 item = wave_first + (u64)sub;
    BlockItem it0;
    bool valid = list_item(lv, items, item, it0);
    if (only_wide) {
        if (valid) valid = (results[item].words & kWideFlag) != 0;
        if (!__any(valid)) { __syncthreads(); return; }
    }
    const u64 grp = item >> 6;
    const int il = (int)(item & 63);
    int qn = 0, tn = 0;
    if (valid) { qn = it0.qn; tn = it0.tn; }
    const int nblk = (qn + 63) >> 6, W = nblk * 64 - qn;
    const bool have = valid && b < nblk;
    const bool is_last = have && b == nblk - 1;
    const u64* fr = frag + grp * FW * 64 + il;
    u64 nlo = 0, nhi = 0;
    if (have) { nlo = fr[(u64)b * 64]; nhi = fr[(u64)(NW + b) * 64]; }
    const u64 pad = (is_last && W > 0) ? (~0ULL << ((64 - W) & 63)) : 0ULL;
    // target words of the wave's blocks staged in LDS: a lane needs a new 32-column word every 32
    // steps, at a step that differs per lane; served from global memory that read would sit in the same
    // vmcnt queue as the NW band stores and stall every step on them
    // staged as two bit-planes per 32 columns (low / high bit of the base code in the low / high half): a lane then
    // gets its column's two symbol masks with one v_bfe_i32 each instead of shift + mask + 64-bit shift + extends
    if (valid) for (int w = b; w < TW; w += G) {
        const u64 x = (w * 32 < tn) ? fr[(u64)(2 * NW + w) * 64] : 0ULL;
        t_lds[sub][w] = even_bits(x) | (even_bits(x >> 1) << 32);
    }
    __syncthreads();
    const u64* tw = t_lds[sub];
    if (!SINGLE && G == 8 && NW == kWordsA && TW == kTWordsA && filter && fast_ok) {
        // every block of the wave a full 512 x 512 one (most waves of the big rounds): the predicate-free path
        if (__all(valid && qn == kOcaBlockSize && tn == kOcaBlockSize)) {
            const u64 ugrp = (u64)__builtin_amdgcn_readfirstlane((int)grp);      // the 8 items of a wave share a slab
            myers_fast_full<NW, TW>(lane, tw, nlo, nhi, slab_records(slabs + (size_t)ugrp * slab_bytes), il, error, valid, results + item, stats, fast_nostore);
            return;
        }
    }
    const u32 nlo_l = (u32)nlo, nlo_h = (u32)(nlo >> 32), nhi_l = (u32)nhi, nhi_h = (u32)(nhi >> 32);
    const u32 pad_l = (u32)pad, pad_h = (u32)(pad >> 32);
    u64 tcur = 0;
    // Eq of column c for this lane's 64 rows: rows whose base code equals the column's
    auto eq_of = [&](int c) -> u64 {
        const u32 ma = (u32)__builtin_amdgcn_sbfe((int)(u32)tcur, (u32)c & 31u, 1u);           // all ones iff bit 0 of the code
        const u32 mb = (u32)__builtin_amdgcn_sbfe((int)(u32)(tcur >> 32), (u32)c & 31u, 1u);   // all ones iff bit 1
        const u32 el = ((nlo_l ^ ma) & (nhi_l ^ mb)) | pad_l, eh = ((nlo_h ^ ma) & (nhi_h ^ mb)) | pad_h;
        return ((u64)eh << 32) | el;
    };
    ulonglong2* rec = slab_records(slabs + (size_t)grp * slab_bytes);

    // wave-uniform trip count of the SHW wavefront
    int steps = valid ? tn + nblk - 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(steps, o); steps = x > steps ? x : steps; }

    // ------------------------------------------------------------------ SHW (edlib_ex.c:108-223, band = everything)
    int k = (int)((double)(qn < tn ? qn : tn) * error * 1.1);
    u64 P = ~0ULL, M = 0ULL;
    int S = (b + 1) * 64, best = -1, end0 = -1, hout = 1;
    for (int s = 0; s < steps; ++s) {
        const int c = s - b;
        int hin = lane_below<G>(hout);
        if (b == 0) hin = 1;
        if (have && (u32)c < (u32)tn) {
            if ((c & 31) == 0) tcur = tw[c >> 5];
            const u64 eq = eq_of(c);
            u64 rA, rB;
            hout = SINGLE ? advance_dev<true>(P, M, eq, hin, rA, rB) : advance_dev<false>(P, M, eq, hin, rA, rB);
            S += hout;
            if (SINGLE) rec[rec_pos<NW>(c, b, il)] = make_ulonglong2(rA, rB);
            if (is_last && S <= k && (best == -1 || S <= best)) {
                if (S != best) { best = S; k = best; end0 = c - W; }
            }
        }
    }
    if (is_last && W > 0) {          // edlib_ex.c:205-219
        int score = S;
        for (int i = 0; i < W; ++i) {
            if (P & (kHighBit >> i)) --score;
            if (M & (kHighBit >> i)) ++score;
            if (score <= k && (best == -1 || score <= best)) {
                if (score != best) { k = best = score; end0 = tn - W + i; }
            }
        }
    }
    // the owner of the last word broadcasts (best, end0) to its group
    const int owner = sub * G + (nblk > 0 ? nblk - 1 : 0);
    best = __shfl(best, owner); end0 = __shfl(end0, owner);
    if (!valid) best = -1;

    // ------------------------------------------------------------------ NW on target[0..end0] with k = best (edlib_ex.c:226-370)
    const int tn2 = end0 + 1;
    int err = 0;
    if (best >= 0) { int ad = tn2 - qn; if (ad < 0) ad = -ad; if (best < ad) err = 1; }
    const bool go = !SINGLE && have && best >= 0 && !err;
    steps = go ? tn2 + nblk - 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(steps, o); steps = x > steps ? x : steps; }
    P = ~0ULL; M = 0ULL; S = (b + 1) * 64; hout = 1;
    u32 kept = 0;
    for (int s = 0; s < steps; ++s) {
        const int c = s - b;
        int hin = lane_below<G>(hout);
        if (b == 0) hin = 1;
        if (go && (u32)c < (u32)tn2) {
            if ((c & 31) == 0) tcur = tw[c >> 5];
            const u64 eq = eq_of(c);
            u64 rA, rB;
            hout = advance_dev<true>(P, M, eq, hin, rA, rB);
            S += hout;
            // store the word only if it can hold a cell of an alignment of cost <= best that still reaches
            // the end: the reference's own per-word band tests (edlib_ex.c:311-325) with k = best.  The
            // traceback never stands on a cell of a dropped word (every cell it visits lies on such an alignment).
            const int rb = (b + 1) * 64 - 1;
            const bool drop = S >= best + 64 || rb > best - S + 2 * 64 - 2 - tn2 + c + qn + 1 || rb < S - best - tn2 + qn + c;
            if (!drop || !filter) { rec[rec_pos<NW>(c, b, il)] = make_ulonglong2(rA, rB); ++kept; }
        }
    }
    if (!SINGLE) {
        for (int o = 32; o > 0; o >>= 1) kept += (u32)__shfl_xor((int)kept, o);
        if (lane == 0 && kept) stat_add(stats, 2, (unsigned long long)kept);
    }
    if (is_last) {
        if (!SINGLE && best >= 0 && !err) {
            int cs = S;
            if (W > 0) cs = S - popc64(P >> ((64 - W) & 63)) + popc64(M >> ((64 - W) & 63));
            if (cs != best) err = 2;
        }
        BlockResult br; br.dist = err ? -1 : best; br.endc = end0; br.err = err;
        br.words = (u32)(nblk * (tn + (!SINGLE && best >= 0 ? tn2 : 0)));
        if (!only_wide) results[item] = br;
        stat_add(stats, 0, (unsigned long long)br.words); stat_add(stats, 1, (unsigned long long)(qn + tn));
    }
}

// (list A's instantiations need 68 VGPRs as compiled freely: asked for 8 waves per SIMD they fit 64 with 28 bytes of scratch, and
// the issue-bound kernel gains ~1 % from the extra wave; list B's keep what they need)
template <int NW, int TW, int COLS, int G, bool SINGLE = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NW == kWordsA ? 8 : 1, 8)))
k_myers_coop(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, char* __restrict__ slabs, size_t slab_bytes,
             double error, BlockResult* __restrict__ results, unsigned long long* __restrict__ stats, u32 epoch, u32 item_base)
{
    constexpr int BPW = 64 / G;
    const ListView lv = list_view(n_host, n_dev, capA);
    const u64 first = (u64)item_base + (u64)blockIdx.x * BPW;
    if (first >= lv.n) return;      // a whole wave beyond the list (grids are sized from an upper bound)
    __shared__ u64 t_lds[BPW][TW];
    myers_coop_wave<NW, TW, COLS, G, SINGLE>(lv, items, frag, slabs, slab_bytes, error, results, stats, epoch, first, t_lds, (int)threadIdx.x);
}

// The list-A DP kernel of the big rounds.  One workgroup of two waves = 16 consecutive work items.  All 16 full 512 x 512 blocks
// (with the list filled from both ends: almost every group of the front part): SHW with 8 lanes per block, the two waves side by
// side, then NW with 4 lanes per block on one wave (ext_fast16.h).  Anything else: each wave does its 8 items the general way
// (myers_coop_wave, which itself has the 8-lane fast path for 8 full blocks).
template <int NW, int TW, int COLS>
__global__ void __launch_bounds__(128)
k_myers_a16(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, char* __restrict__ slabs, size_t slab_bytes,
            double error, BlockResult* __restrict__ results, unsigned long long* __restrict__ stats, u32 epoch)
{
    constexpr int FW = 2 * NW + TW;
    const ListView lv = list_view(n_host, n_dev, capA);
    const u64 first = (u64)blockIdx.x * 16;
    if (first >= lv.n) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ u64 tl[16][TW];
    __shared__ int res[16][2];
    BlockItem it;
    bool full = list_item(lv, items, first + (u64)(lane & 15), it);
    if (full) full = it.qn == kOcaBlockSize && it.tn == kOcaBlockSize;
    const bool f16 = ((epoch >> 27) & 1u) != 0;
    if (f16 && __all(full)) {
        const u64 grp = first >> 6;
        const int il0 = (int)(first & 63);
        myers_fast16<NW, TW>(lane, frag + grp * FW * 64 + il0, slabs + (size_t)grp * slab_bytes, il0, error, results + first, stats, tl, res, ((epoch >> 28) & 1u) != 0);
        return;
    }
    myers_coop_wave<NW, TW, COLS, 8, false>(lv, items, frag, slabs, slab_bytes, error, results, stats, epoch, first + 8ull * wv, tl + 8 * wv, lane);
}

// maximum over the active lanes of a value below 8192 (bit by bit with ballots: exited lanes do not take part)
NECAT_D int wave_max_u11(int v)
{
    int best = 0;
    bool in = true;
    for (int bit = 12; bit >= 0; --bit) {
        const bool has = in && ((v >> bit) & 1);
        if (__ballot(has)) { best |= 1 << bit; in = has; }
    }
    return best;
}

struct OpsWriter {
    u8* ops; int cap; int overflow; bool store;
    TailScan ts;
    NECAT_D void push(int op)
    {
        if (store) { if (ts.n < cap) ops[(size_t)ts.n * 64] = (u8)op; else overflow = 1; }
        tail_push(ts, op);
    }
};
struct OpsSink {        // walk_block's op store: op number i of the lane's block at ops[i * 64]
    u8* ops; int cap; int overflow; bool store;
    NECAT_D bool storing() const { return store; }
    NECAT_D void put(int i, int op) { if (i < cap) ops[(size_t)i * 64] = (u8)op; else overflow = 1; }
};
struct OpsReader {
    const u8* ops;
    NECAT_D int operator()(int j) const { return ops[(size_t)j * 64]; }
};
template <int NW>
struct SameReader {   // query fragment element i == target fragment element i ?
    const u64* fr;
    NECAT_D bool operator()(int i) const
    {
        const u64 nlo = fr[(u64)(i >> 6) * 64], nhi = fr[(u64)(NW + (i >> 6)) * 64];
        const int q = (int)((~nlo >> (i & 63)) & 1) | ((int)((~nhi >> (i & 63)) & 1) << 1);
        const u64 tw = fr[(u64)(2 * NW + (i >> 5)) * 64];
        return q == (int)((tw >> ((i & 31) * 2)) & 3);
    }
};

// EXPORT = false: fold the block into its ExtTask.  EXPORT = true (batch API): keep the ops.
// WALK: 0 = traceback_block (the reference formulation, the default), 1 = walk_block, 2 = walk_block without record prefetch
// (a template parameter, not a run-time switch: the two walks in one kernel cost the faster one its registers)
// WAVES: waves per workgroup (1, or 4 for the finishing launches of the big lists: one reservation per workgroup and list, ext_append_block_wg)
template <int NW, int TW, int COLS, int MAXOPS, bool EXPORT, int WALK = 0, int BLOCK = kOcaBlockSize, bool ONE_LIST = false, int WAVES = 1>
__global__ void __launch_bounds__(64 * WAVES)
k_traceback(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, const char* __restrict__ slabs, size_t slab_bytes,
            const BlockResult* __restrict__ results, u8* __restrict__ ops_pool, ExtTask* __restrict__ tasks, int tail_match_len,
            i32* __restrict__ n_ops_out, int* __restrict__ err_flag, ExtLists next, u32 epoch, u32 item_base = 0, const WalkOut* __restrict__ wout = nullptr)
{
    constexpr int FW = 2 * NW + TW;
    const ListView lv = list_view(n_host, n_dev, capA);
    const u32 grp = blockIdx.x * WAVES + (threadIdx.x >> 6) + (item_base >> 6);      // item_base: a multiple of 64 (a list handled in several launches: bounded band pool)
    const int lane = threadIdx.x & 63;
    const u64 item = (u64)grp * 64 + lane;
    BlockItem it;
    ExtTask t;
    // (every thread reaches the list append below: the workgroup form of it has barriers)
    auto block_of_lane = [&]() -> bool {
    if (!list_item(lv, items, item, it)) return false;
    const BlockResult br = results[item];
    // WALK == 5: the walks were done by k_rcwalk4 (the wide blocks it left out are walked by an `only wide` launch: epoch bit 25)
    // (epoch bit 27: every block of the list, not only the full ones at the front of a two-ended list A)
    if (WALK == 5 && ((item >= lv.nf && !((epoch >> 27) & 1u)) || (br.words & kWideFlag))) return false;
    if (WALK != 5 && ((epoch >> 26) & 1u) && item < lv.nf16) return false;
    if (WALK != 5 && (((epoch >> 25) & 1u) || (((epoch >> 24) & 1u) && item < lv.nf16)) && !(br.words & kWideFlag)) return false;
    if (br.err) atomicExch(err_flag, 10 + br.err);
    OpsWriter ow; ow.ops = ops_pool + (size_t)grp * MAXOPS * 64 + lane; ow.cap = MAXOPS; ow.overflow = 0; ow.store = true;
    int done = 0;
    if (!EXPORT) {
        t = tasks[it.task];
        done = ext_block_done(t, br.dist, br.endc);
        // the op list is only replayed until the stream's first run of 8 matches - unless the caller keeps the columns
        ow.store = !t.found || next.task_ops != nullptr;
    }
    tail_init(ow.ts, (EXPORT || !done) ? kOcaMatCnt : tail_match_len);
    if (WALK == 5) {
        if (br.dist >= 0) {
            const WalkOut o = wout[item];
            ow.ts.n = o.n; ow.ts.nq = it.qn; ow.ts.nt = br.endc + 1; ow.ts.nmat = o.nmat; ow.ts.m = o.m; ow.ts.hit = o.hit;
            ow.ts.acnt = o.acnt; ow.ts.qcnt = o.qcnt; ow.ts.tcnt = o.tcnt; ow.ts.mcnt = o.mcnt;
        }
    } else
    if (br.dist >= 0) {
        MatReader<NW> mr;
        mr.base = slab_records(const_cast<char*>(slabs) + (size_t)grp * slab_bytes); mr.lane = lane; mr.init();
        mr.no_prefetch = WALK == 2;
        mr.nt = WALK >= 3;
        if (WALK == 0 || WALK == 3) traceback_block(it.qn, br.endc + 1, mr, ow);
        else {
            OpsSink sk; sk.ops = ow.ops; sk.cap = ow.cap; sk.overflow = 0; sk.store = ow.store;
            walk_block(it.qn, br.endc + 1, mr, sk, ow.ts);
            ow.overflow = sk.overflow;
        }
        if (ow.overflow) atomicExch(err_flag, 20);
    }
    if (EXPORT) { n_ops_out[item] = ow.ts.n; return false; }
    OpsReader rd; rd.ops = ow.ops;
    SameReader<NW> same; same.fr = frag + (u64)grp * FW * 64 + lane;
    const int stream_at = t.phase == 1 ? t.s_lto : 0;     // where the block's stream starts in the task's column region
    const ExtKept kept = ext_finish_block(t, br.dist, br.endc, done, ow.ts, rd, same);
    if (next.task_ops) {
        // The kept columns join the task's stream, packed 2 bits per column (32 per 64-bit word).  The op pool is
        // lane-interleaved by op index, and op r of a lane is forward column nops - 1 - r: the wave walks the pool
        // rows from the highest one down, every lane reading the SAME row (one 64-byte line per step) and packing
        // its own op while it is inside its kept range; a word goes out once it is full.
        u64* reg = reinterpret_cast<u64*>(next.task_ops + t.ops_base);
        const u32 pos = (u32)(stream_at + kept.at);
        u32 w = pos >> 5; int sh = (int)(pos & 31) * 2;
        u64 acc = sh ? (reg[w] & ((1ULL << sh) - 1)) : 0;       // the columns of earlier blocks in the same word
        const int nops = ow.ts.n;
        if (kept.exact) {
            for (int f = 0; f < kept.cols; ++f) { sh += 2; if (sh == 64) { reg[w++] = acc; acc = 0; sh = 0; } }
        } else {
            const int mine = kept.cols > 0 ? nops : 0;          // rows [nops - cols, nops) hold forward [0, cols)
            const int lo = nops - kept.cols;
            // (32 rows per trip, loaded before any is used: the loads are independent - the loop used to be one memory round trip per row)
            constexpr int kRows = 32;
            for (int r = wave_max_u11(mine) - 1; r >= 0; r -= kRows) {
                u32 b[kRows];
#pragma unroll
                for (int k = 0; k < kRows; ++k) { const int rr = r - k; b[k] = (rr < mine && rr >= lo) ? (u32)ow.ops[(size_t)rr * 64] : 0u; }
#pragma unroll
                for (int k = 0; k < kRows; ++k) {
                    const int rr = r - k;
                    if (rr < mine && rr >= lo) {
                        acc |= (u64)b[k] << sh; sh += 2;
                        if (sh == 64) { reg[w++] = acc; acc = 0; sh = 0; }
                    }
                }
            }
        }
        if (sh) reg[w] = acc;                                   // the bits above `sh` are zero: the next block ORs into them
    }
    const bool go_ = ext_plan<BLOCK>(t);      // schedule the candidate's next block for the next round (or finish it)
    tasks[it.task] = t;
    return go_;
    };
    const bool go = block_of_lane();
    if (EXPORT) return;
    if (WAVES > 1) ext_append_block_wg<BLOCK, ONE_LIST, WAVES>(t, go ? (u32)it.task : 0u, go, next);
    else ext_append_block<BLOCK, ONE_LIST>(t, go ? (u32)it.task : 0u, go, next);
}

// ---- final records: pm_worker.c:56-80 (M4 fields), oc_aligner.c:419-450 (coordinates, identity) ----
__global__ void __launch_bounds__(256)
k_ext_result(const ExtTask* __restrict__ tasks, u32 n, const necat_candidate* __restrict__ cands,
             int min_align, necat_m4* __restrict__ m4, u8* __restrict__ ok, int window)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ExtTask t = tasks[i];
    const u32 ci = (u32)t.cand;                  // results go to the candidate's own slot, whatever the batch order
    const necat_candidate c = cands[ci];
    necat_m4 m;
    m.qid = c.qid; m.qdir = c.qdir; m.qoff = (u64)t.r_qoff; m.qend = (u64)t.r_qend; m.qext = c.qoff; m.qsize = c.qsize;
    m.sid = c.sid; m.sdir = 0; m.soff = (u64)t.r_toff; m.send = (u64)t.r_tend; m.sext = c.soff; m.ssize = c.ssize;
    m.ident_perc = t.r_cols ? 100.0 * (double)t.r_mat / (double)t.r_cols : 0.0;
    m.vscore = c.score; m._pad = 0;
    if (window) {          // rm_worker.c:148-149: back to coordinates of the whole reference sequence
        i64 from, to, woff;
        rm_window((i64)c.qoff, (i64)c.qsize, (i64)c.soff, (i64)c.ssize, &from, &to, &woff);
        m.soff += (u64)from; m.send += (u64)from;
    }
    if (m.qdir == 1) { const u64 qo = m.qsize - m.qend, qe = m.qsize - m.qoff; m.qoff = qo; m.qend = qe; }
    m4[ci] = m;
    ok[ci] = t.r_cols >= min_align ? 1 : 0;
}

// necat_onc_align_batch: per-candidate results in strand coordinates (what onc_align leaves in OcAlignData)
__global__ void __launch_bounds__(256)
k_ext_alignment(const ExtTask* __restrict__ tasks, u32 n, u32 cand_base, int min_align, necat_alignment* __restrict__ out, u32* __restrict__ lens)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ExtTask t = tasks[i];
    necat_alignment a;
    a.ok = t.r_cols >= min_align ? 1 : 0;
    a.qoff = t.r_qoff; a.qend = t.r_qend; a.toff = t.r_toff; a.tend = t.r_tend; a.align_size = t.r_cols;
    a.ident_perc = t.r_cols ? 100.0 * (double)t.r_mat / (double)t.r_cols : 0.0;
    out[cand_base + i] = a;
    lens[i] = (u32)t.r_cols;
}

// final alignment of task i = its left stream [s_lfrom, s_lto) reversed, then its right stream
// [s_rfrom, s_rto) (oc_aligner.c:358-366, :404-428); one wave per task, coalesced
// Columns are 2 bits each, 32 per 64-bit word, column j of an alignment at bits 2 (j & 31) of its word j >> 5.
__global__ void __launch_bounds__(256)
k_ext_strings(const ExtTask* __restrict__ tasks, u32 n, const u8* __restrict__ task_ops, const u64* __restrict__ out_off, u64* __restrict__ out)
{
    const u32 wave = (u32)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= n) return;
    const ExtTask t = tasks[wave];
    const u64* reg = reinterpret_cast<const u64*>(task_ops + t.ops_base);
    u64* dst = out + out_off[wave];
    const int nl = t.s_lto - t.s_lfrom, nr = t.s_rto - t.s_rfrom;
    const int nw = (nl + nr + 31) >> 5;
    for (int w = lane; w < nw; w += 64) {
        u64 acc = 0;
        const int j0 = w * 32;
        if (j0 >= nl && j0 + 32 <= nl + nr) {
            // inside the right stream: 32 consecutive columns = 64 consecutive bits of the region
            const u32 src = (u32)(t.s_lto + t.s_rfrom + (j0 - nl));
            const u32 sw = src >> 5; const int sh = (int)(src & 31) * 2;
            acc = reg[sw] >> sh;
            if (sh) acc |= reg[sw + 1] << (64 - sh);
        } else {
            for (int c = 0; c < 32; ++c) {
                const int j = j0 + c;
                if (j >= nl + nr) break;
                const u32 src = j < nl ? (u32)(t.s_lto - 1 - j) : (u32)(t.s_lto + t.s_rfrom + (j - nl));
                acc |= ((reg[src >> 5] >> ((src & 31) * 2)) & 3ULL) << (2 * c);
            }
        }
        dst[w] = acc;
    }
}

// extend_candidates' containment rule (pm_worker.c:44, map_aux.c:4-20), one lane per query read:
// a candidate whose anchor lies inside an already ACCEPTED record of the same (qdir, sid) is dropped.
// Aligning every candidate first and filtering afterwards is equivalent because acceptance of a
// candidate depends only on its own alignment.
// One WAVE per read (a lane per read walked ~ 10 x 10 dependent global loads: 0.45 ms at E. coli size): lane j holds the
// fields of the group's record j that the test reads (records beyond 64 per group: the lanes take several), the candidates are
// decided in order, each against the accepted ones before it by one ballot.
__global__ void __launch_bounds__(256)
k_m4_filter(const necat_candidate* __restrict__ cands, const u64* __restrict__ group_off, u32 n_groups,
            const necat_m4* __restrict__ m4, u8* __restrict__ ok, necat_m4* __restrict__ out, u32* __restrict__ out_count)
{
    const u32 g = (u32)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = (int)(threadIdx.x & 63);
    if (g >= n_groups) return;
    const u64 lo = group_off[g], hi = group_off[g + 1];
    const u32 n = (u32)(hi - lo);
    if (n <= 64) {
        // lane j: record j and candidate j of the group
        int mq = 0, ms = 0, okj = 0; u64 mqoff = 0, mqend = 0, msoff = 0, msend = 0;
        int cq = 0, cs = 0; u64 cqoff = 0, csoff = 0;
        if ((u32)lane < n) {
            const necat_m4& m = m4[lo + lane];
            mq = m.qdir; ms = m.sid; mqoff = m.qoff; mqend = m.qend; msoff = m.soff; msend = m.send;
            const necat_candidate& c = cands[lo + lane];
            cq = c.qdir; cs = c.sid; cqoff = (u64)c.qoff; csoff = (u64)c.soff;
            okj = ok[lo + lane];
        }
        u64 accepted = 0;                     // lanes whose record is accepted so far
        for (u32 i = 0; i < n; ++i) {
            const int iq = __shfl(cq, (int)i), is = __shfl(cs, (int)i);
            const u64 iqoff = __shfl(cqoff, (int)i), isoff = __shfl(csoff, (int)i);
            const bool hit = ((accepted >> lane) & 1ULL) && iq == mq && is == ms && iqoff >= mqoff && iqoff <= mqend && isoff >= msoff && isoff <= msend;
            const bool contained = __ballot(hit) != 0ULL;
            const int oki = __shfl(okj, (int)i);
            if (!contained && oki == 1) accepted |= 1ULL << i;
        }
        if ((u32)lane < n) {
            const bool acc = (accepted >> lane) & 1ULL;
            ok[lo + lane] = acc ? 2 : 0;
        }
        const u32 cnt = (u32)popc64(accepted);
        u32 base = 0;
        if (lane == 0 && cnt) base = atomicAdd(out_count, cnt);
        base = __shfl(base, 0);
        if ((accepted >> lane) & 1ULL) out[base + (u32)popc64(accepted & ((1ULL << lane) - 1ULL))] = m4[lo + lane];
        return;
    }
    if (lane != 0) return;                    // a read with more than 64 candidates (-n > 64 and a repeat-rich read): the sequential form
    for (u64 i = lo; i < hi; ++i) {
        const necat_candidate c = cands[i];
        bool contained = false;
        for (u64 j = lo; j < i && !contained; ++j) {
            if (ok[j] != 2) continue;
            const necat_m4& m = m4[j];
            contained = c.qdir == m.qdir && c.sid == m.sid && c.qoff >= m.qoff && c.qoff <= m.qend &&
                        c.soff >= m.soff && c.soff <= m.send;
        }
        if (!contained && ok[i] == 1) { ok[i] = 2; out[atomicAdd(out_count, 1u)] = m4[i]; }
        else ok[i] = 0;
    }
}

}  // namespace necat
