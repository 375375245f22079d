// seed_kernels.h - __global__ shells around seed_core.h.
//   k_seed_hits    : one wave per query read; lanes stride over the sampled k-mers of both strands and
//                    sum their occurrence counts.  Random 8-byte gathers into kmer_stats (HBM/L2
//                    latency bound; SURVEY.md 8d "seeding" row).  The totals bound every per-read
//                    scratch size, so the later kernels never reallocate.
//   k_seed_collect : one lane per (read, strand): collect_seeds on a sparse per-strand block table
//                    (seed_core.h); uniform loop nest, lanes differ only in trip counts.
//   k_seed_eval    : one WAVE per read: the order-dependent walk over the touched blocks, with the
//                    O(n^2) DDF vote and the co-linear gather of every evaluation spread over the lanes.
//                    Reads are visited in descending hit-count order.
//   k_pack_cands   : compaction of the per-read outputs into one array of necat_candidate (global ids).
#pragma once
#include "seed_core.h"

namespace necat {

struct SeedMeta {           // per processed read; strand 0 = FWD, 1 = REV
    u64 ht_off[2];          // entries
    u64 pool_off[2];        // SBlocks
    u64 chain_off[2];       // chain scratch per strand, H + 1 entries each: the strands are evaluated by two waves at the same time
    u64 out_off;            // DevCands: strand 0 writes at out_off, strand 1 at out_off + out_cap0; k_seed_finish joins them
    u32 ht_mask[2], pool_cap[2], cs_cap[2], out_cap, out_cap0;
};

// Where the table words of read r's sampled k-mers are kept between k_seed_hits (which fetches them to size the scratch) and
// k_seed_collect_wave (which used to fetch them again: two dependent random loads per k-mer, twice): entry 2 * i + strand of the
// read's stretch, which starts at 2 * (g0 / z + r) - a read of L bases has at most L / z + 1 sampled k-mers, so the stretches of
// consecutive reads never overlap and no per-read offset table is needed.  2 * (nbases / z + nreads + 1) words in all.
NECAT_HD u64 seed_kst_base(u64 g0, u32 r, int z) { return 2 * (g0 / (u64)z + (u64)r); }

__global__ void __launch_bounds__(256)
k_seed_hits(DevVolume reads, IndexView index, int k, int z, u32 read_lo, u32 read_hi, u32* __restrict__ hits, u64* __restrict__ kst)
{
    const u32 wave = (u32)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    const u32 r = read_lo + wave;
    if (r >= read_hi) return;
    const u64 g0 = reads.seq_off[r];
    const int L = (int)(reads.seq_off[r + 1] - g0);
    const int nk = L >= k ? (L - k) / z + 1 : 0;
    u32 hf = 0, hr = 0;
    for (int i = lane; i < nk; i += 64) {
        const int pos = i * z;
        const u64 xf = load32_dir(reads.bases, (i64)g0 + pos, +1, 0);
        const u64 xr = load32_dir(reads.bases, (i64)g0 + L - 1 - pos, -1, 1);
        const u64 sf = index.lookup(rev2(xf) >> (64 - 2 * k)), sr = index.lookup(rev2(xr) >> (64 - 2 * k));
        hf += (u32)(sf >> kOffsetBits);
        hr += (u32)(sr >> kOffsetBits);
        if (kst) { u64* dst = kst + seed_kst_base(g0, r, z) + 2 * (u64)i; dst[0] = sf; dst[1] = sr; }
    }
    for (int o = 32; o > 0; o >>= 1) { hf += __shfl_down(hf, o); hr += __shfl_down(hr, o); }
    if (lane == 0) { hits[2 * (u64)r] = hf; hits[2 * (u64)r + 1] = hr; }
}

struct SeedArenas {
    u64* ht; SBlock* pool;
    u64* cs; i32* f; i32* p; i32* t; i32* v; u64* u; DevCand* lcan;
    DevCand* out;
};

NECAT_D SeedScratch seed_scratch(const SeedArenas& A, const SeedMeta& m, int strand)
{
    SeedScratch S;
    S.ht = A.ht + m.ht_off[strand]; S.ht_mask = m.ht_mask[strand];
    S.pool = A.pool + m.pool_off[strand]; S.pool_cap = m.pool_cap[strand];
    const u64 co = m.chain_off[strand];
    S.cs = A.cs + co; S.f = A.f + co; S.p = A.p + co; S.t = A.t + co;
    S.v = A.v + co; S.u = A.u + co; S.lcan = A.lcan + co; S.cs_cap = m.cs_cap[strand];
    S.out = A.out + m.out_off + (strand ? m.out_cap0 : 0u); S.out_cap = strand ? m.out_cap - m.out_cap0 : m.out_cap0;
    return S;
}

// Seed collection: one lane per (read, strand); every lane runs the same loop nest (sampled k-mers x
// their occurrence lists), so lanes diverge only in trip counts.
#if NECAT_XCHECK          // the lane-per-strand collection k_seed_collect_wave replaced: cross-check build only (necat_hip.hip)
__global__ void __launch_bounds__(64)
k_seed_collect(DevVolume ref, DevVolume reads, IndexView index, const u64* __restrict__ offset_list,
               SeedParams P, const u32* __restrict__ order, const SeedMeta* __restrict__ meta, u32 n,
               SeedArenas A, i32* __restrict__ nblk_out, int* __restrict__ err_flag)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * n) return;
    const u32 i = t >> 1;
    const int strand = (int)(t & 1);
    SeedScratch S = seed_scratch(A, meta[i], strand);
    const int nb = seed_collect_strand(ref, index, offset_list, reads, (int)order[i], strand, P, S);
    if (nb < 0) atomicExch(err_flag, 1);
    nblk_out[t] = nb < 0 ? 0 : nb;
}
#endif

// Seed collection, one WAVE per (read, strand).  collect_seeds (word_finder.c:107-139) is a strictly
// ordered walk - sampled k-mers in read order, each k-mer's occurrences in ascending offset order - whose
// order shows in the results (first-touch order of the blocks, the first 40 seeds of a block, the stale
// pair score).  The walk is flattened into one seed sequence (k-mers of a group of 64 x their occurrence
// lists, via a prefix sum in LDS) and taken 64 consecutive seeds at a time; inside such a chunk lane
// order IS sequence order, so everything the sequential walk would have seen is a count over lower lanes:
//   * a seed is dropped when the previous occurrence of its k-mer falls into the same block
//     (the last_kmer_id test of fill_one_seed, word_finder.c:93; occurrence lists are ascending);
//   * rank = earlier seeds of the chunk in the same block  -> its slot is score + rank, kept while < 40;
//   * new blocks get pool slots in lane order (= first-touch order);
//   * the stale score written with a block's last kept seed uses the neighbour's score + the neighbour's
//     kept seeds from lower lanes.
// One chunk costs a handful of dependent memory round trips for 64 seeds instead of 64 x that per lane.
__global__ void __launch_bounds__(64)
k_seed_collect_wave(DevVolume ref, DevVolume reads, IndexView index, const u64* __restrict__ offset_list,
                    SeedParams P, const u32* __restrict__ order, const SeedMeta* __restrict__ meta, u32 n,
                    SeedArenas A, i32* __restrict__ nblk_out, int* __restrict__ err_flag, const u64* __restrict__ kst)
{
    __shared__ u32 s_pre[65];
    __shared__ u64 s_list[64];
    const u32 t = blockIdx.x;
    if (t >= 2 * n) return;
    const u32 i = t >> 1;
    const int strand = (int)(t & 1);
    const int lane = threadIdx.x;
    const u64 below = (1ULL << lane) - 1ULL;
    SeedScratch S = seed_scratch(A, meta[i], strand);
    const int read_id = (int)order[i];
    const u64 q_goff = reads.seq_off[read_id];
    const int L = (int)(reads.seq_off[read_id + 1] - q_goff);
    u64 soff_max = ~0ULL;
    if (P.pairwise) {
        const int gid = read_id + P.read_start_id;
        if (gid >= P.ref_start_id && gid < P.ref_start_id + (int)ref.nseq) soff_max = ref.seq_off[read_id];
    }
    const int k = P.k, z = P.z;
    const double bsd = (double)P.block_size;
    const u64 bs = (u64)P.block_size;
    const int nk = L >= k ? (L - k) / z + 1 : 0;
    int nblk = 0;
    bool failed = false;
    for (int kbase = 0; kbase < nk && !failed; kbase += 64) {
        // ---- the group's k-mers: hash, occurrence list, occurrences below soff_max
        const int kj = kbase + lane;
        u32 cnt = 0; u64 lst = 0;
        if (kj < nk) {
            u64 st;
            if (kst) st = kst[seed_kst_base(q_goff, (u32)read_id, z) + 2 * (u64)kj + (u64)strand];      // fetched by k_seed_hits
            else {
                const int pos = kj * z;
                const u64 x = strand == 0 ? load32_dir(reads.bases, (i64)q_goff + pos, +1, 0)
                                          : load32_dir(reads.bases, (i64)q_goff + L - 1 - pos, -1, 1);
                st = index.lookup(rev2(x) >> (64 - 2 * k));
            }
            cnt = (u32)(st >> kOffsetBits); lst = st & kOffsetMask;
            if (cnt && soff_max != ~0ULL) {
                u32 lo = 0, hi = cnt;
                while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (offset_list[lst + mid] < soff_max) lo = mid + 1; else hi = mid; }
                cnt = lo;
            }
        }
        u32 inc = cnt;
        for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        __syncthreads();                       // the previous group's chunks are done with s_pre / s_list
        s_pre[lane] = inc - cnt; s_list[lane] = lst;
        if (lane == 63) s_pre[64] = inc;
        __syncthreads();
        const u32 T = s_pre[64];
        // the seed of lane `lane` in the chunk that starts at c0: its k-mer (the last j with s_pre[j] <= sq), its offset and the
        // previous occurrence of the k-mer.  Fetched one chunk ahead: these loads depend on nothing the chunk loop produces, and
        // the loop is a chain of dependent memory round trips.
        auto fetch = [&](u32 c0, int& j, u32& kk, u64& off, u64& offp) -> bool {
            const u32 sq = c0 + (u32)lane;
            j = 0;
            for (int step = 32; step > 0; step >>= 1) if (s_pre[j + step] <= sq) j += step;
            kk = sq - s_pre[j];
            off = 0; offp = 0;
            if (sq >= T) return false;
            const u64 lbase = s_list[j];
            off = offset_list[lbase + kk];
            if (kk > 0) offp = offset_list[lbase + kk - 1];
            return true;
        };
        int nj = 0; u32 nkk = 0; u64 noff = 0, noffp = 0;
        bool nvalid = T ? fetch(0, nj, nkk, noff, noffp) : false;
        for (u32 c0 = 0; c0 < T; c0 += 64) {
            const int j = nj; const u32 kk = nkk; const u64 off = noff, offp = noffp;
            const bool valid = nvalid;
            if (c0 + 64 < T) nvalid = fetch(c0 + 64, nj, nkk, noff, noffp);
            i32 blk = -2; int boff = 0; bool cand = false;
            if (valid) {
                u64 q = (u64)((double)off / bsd);
                if (q * bs > off) --q; else if ((q + 1) * bs <= off) ++q;
                blk = (i32)q; boff = (int)(off - q * bs);
                cand = true;
                if (kk > 0) cand = !(offp >= q * bs);                 // offp < off: same block iff offp >= block start
            }
            const i32 kmer_id = kbase + j + 1;
            // ---- counts over the lower / higher lanes of the chunk
            const u64 cmask = __ballot(cand);
            // one step per DISTINCT block of the chunk (consecutive k-mers of a read hit the same few blocks of an overlapping
            // read), not one per lane
            int rank = 0, first = lane, later = 0, cprev = 0;
            for (u64 todo = cmask; todo;) {
                const int l = ctz64(todo);
                const i32 bl = __builtin_amdgcn_readlane(blk, l);
                const bool mine = cand && blk == bl;
                const u64 same = __ballot(mine);                      // the candidates of block bl; l is the lowest of them
                if (mine) { rank = popc64(same & below); later = popc64(same >> lane) - 1; first = l; }
                if (cand && blk == bl + 1) cprev = popc64(same & below);
                todo &= ~same;
            }
            // ---- block lookup / creation by the first seed of every block
            const bool is_first = cand && rank == 0;
            // (entries are inserted with L2 atomics: probed with agent-scope loads so that a stale L1 line is never read)
            // the first seed of a block looks the block up; every candidate looks up the LEFT neighbour too (its score goes into
            // the stale pair score written with the block's last kept seed): both first loads fly together.  A neighbour created
            // by this very chunk may be missed here - it has no seeds yet, which is what a miss counts as.
            i32 idx = -1, ip = -1; u32 h = 0;
            {
                u32 hp = 0;
                const bool want_own = is_first, want_prev = cand && blk > 0;
                u64 e_own = kHtEmpty, e_prev = kHtEmpty;
                if (want_own) { h = ht_hash(blk, S.ht_mask); e_own = __hip_atomic_load(&S.ht[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                if (want_prev) { hp = ht_hash(blk - 1, S.ht_mask); e_prev = __hip_atomic_load(&S.ht[hp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                if (want_own) {
                    while (e_own != kHtEmpty && (i32)(u32)e_own != blk) { h = (h + 1) & S.ht_mask; e_own = __hip_atomic_load(&S.ht[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                    idx = e_own == kHtEmpty ? -1 : (i32)(u32)(e_own >> 32);
                }
                if (want_prev) {
                    while (e_prev != kHtEmpty && (i32)(u32)e_prev != blk - 1) { hp = (hp + 1) & S.ht_mask; e_prev = __hip_atomic_load(&S.ht[hp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                    ip = e_prev == kHtEmpty ? -1 : (i32)(u32)(e_prev >> 32);
                }
            }
            const bool create = is_first && idx < 0;
            const u64 crm = __ballot(create);
            const int ncreate = popc64(crm);
            if ((u32)(nblk + ncreate) > S.pool_cap) { failed = true; break; }
            if (create) {
                idx = nblk + popc64(crm & below);
                for (;;) {           // the keys of one chunk's creators are distinct: a lost race just moves on
                    const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&S.ht[h]), (unsigned long long)kHtEmpty, (unsigned long long)ht_entry(blk, idx));
                    if (old == (unsigned long long)kHtEmpty) break;
                    h = (h + 1) & S.ht_mask;
                }
                SBlock* sb = S.pool + idx;
                sb->score = 0; sb->last_kmer_id = -1; sb->block_id = blk; sb->stale = 0; sb->slot = (i32)h;
            }
            nblk += ncreate;
            idx = __shfl(idx, first);
            __syncthreads();                   // new blocks are initialised before anyone reads a score
            int s0 = 0, s0p = 0;
            if (cand) s0 = S.pool[idx].score;
            if (cand && ip >= 0) s0p = (int)S.pool[ip].score;
            const bool acc = cand && s0 + rank < kBlkSeeds;
            const bool is_last = acc && (later == 0 || s0 + rank + 1 >= kBlkSeeds);
            int sprev = 0;
            if (is_last) {
                int room = kBlkSeeds - s0p; if (room < 0) room = 0;
                sprev = s0p + (cprev < room ? cprev : room);
            }
            __syncthreads();                   // every score of the chunk is read before any is written
            if (acc) {
                SBlock* sb = S.pool + idx;
                sb->blk_offset[s0 + rank] = (short)boff;
                sb->kmer_id[s0 + rank] = kmer_id;
                if (is_last) { sb->score = (short)(s0 + rank + 1); sb->last_kmer_id = kmer_id; sb->stale = s0 + rank + 1 + sprev; }
            }
            __syncthreads();
        }
    }
    if (lane == 0) {
        if (failed) atomicExch(err_flag, 1);
        nblk_out[t] = failed ? 0 : nblk;
    }
}

// clear_WordFindData (word_finder.c:40-52) for the whole chunk: the hash slots the strands used go back to empty, so the arena is
// all-empty again when the call ends and the next call need not fill it (a 0xFF fill of the arena was 0.9 ms per pass)
__global__ void __launch_bounds__(64)
k_seed_clear(const SeedMeta* __restrict__ meta, u32 n, SeedArenas A, const i32* __restrict__ nblk_in)
{
    const u32 t = blockIdx.x;
    if (t >= 2 * n) return;
    const SeedScratch S = seed_scratch(A, meta[t >> 1], (int)(t & 1));
    const int nb = nblk_in[t];
    for (int b = (int)threadIdx.x; b < nb; b += 64) S.ht[S.pool[b].slot] = kHtEmpty;
}

constexpr int kLdsChain = 256;
constexpr int kLdsCan = 8;          // chains of one evaluation kept in LDS (more: the global scratch)

// ---- chain DP on all 64 lanes (chain_fill / chain_ends of seed_core.h are the sequential statement of the same thing)
// inclusive prefix max / min over the lanes of a wave, DPP: Kogge-Stone inside the rows of 16, then lane 15 -> next row, lane 31 -> rows 2, 3
#define NECAT_DPP_SCAN(NAME, OP)                                                                                  \
NECAT_D int NAME(int x)                                                                                           \
{                                                                                                                 \
    x = OP(x, __builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false));   /* row_shr:1 (lanes without a source keep x) */ \
    x = OP(x, __builtin_amdgcn_update_dpp(x, x, 0x112, 0xf, 0xf, false));                                         \
    x = OP(x, __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xf, false));                                         \
    x = OP(x, __builtin_amdgcn_update_dpp(x, x, 0x118, 0xf, 0xf, false));                                         \
    x = OP(x, __builtin_amdgcn_update_dpp(x, x, 0x142, 0xa, 0xf, false));   /* row_bcast:15 into rows 1, 3 */      \
    x = OP(x, __builtin_amdgcn_update_dpp(x, x, 0x143, 0xc, 0xf, false));   /* row_bcast:31 into rows 2, 3 */      \
    return x;                                                                                                     \
}
NECAT_D int imax2(int a, int b) { return a > b ? a : b; }
NECAT_D int imin2(int a, int b) { return a < b ? a : b; }
NECAT_DPP_SCAN(wave_prefix_max, imax2)
NECAT_DPP_SCAN(wave_prefix_min, imin2)
// (the scans above rely on "lanes without a source keep x": right for max / min, for the sum those lanes must add nothing)
NECAT_D int wave_prefix_sum(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}
// value of lane `l` (the same l on every lane) / of the lane below (lane 0 gets `first`)
NECAT_D int wave_read(int x, int l) { return __builtin_amdgcn_readlane(x, __builtin_amdgcn_readfirstlane(l)); }
NECAT_D int wave_from_below(int x, int first) { return __builtin_amdgcn_update_dpp(first, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
#undef NECAT_DPP_SCAN

// chain_fill for a workgroup of ONE wave.  The predecessor scan j = i - 1 ... st of seed i runs 64 candidates at a time, lane l
// = the l-th j of the chunk, so lane order is the order of the sequential loop and its running state becomes prefix operations:
//   * "sc > max_f" (a new best, strictly): sc above the prefix maximum of the lanes before (and of the chunks before);
//     the last such lane holds the final best.
//   * "t[j] == i" (j is the predecessor of a j' seen earlier in this scan): every lane stores its mark first; a mark can only
//     come from a HIGHER j, i.e. an earlier lane or chunk, so reading after a barrier sees exactly the sequential loop's marks.
//   * n_skip is a walk floored at zero: + 1 for a marked lane that is no new best, - 1 (if > 0) for a new best.  With S_l the
//     plain prefix sum of the steps, n_skip after lane l = S_l - min(-n_skip_before, min_{m <= l} S_m); the scan stops at the
//     first lane where it exceeds max_skip (always a + 1 lane).  Lanes behind the stop still stored their marks: t[] == i is
//     never tested again once the scan of i is over.
template <class I>
NECAT_D void chain_fill_wave(const u64* cs, I* f, I* p, I* t, I* v, int n, int kmer_size, int lane)
{
    const u64 below = (1ULL << lane) - 1ULL;
    for (int a = lane; a < n; a += 64) { f[a] = 0; p[a] = (I)-1; t[a] = 0; v[a] = 0; }
    __syncthreads();
    int st = 0;
    for (int i = 0; i < n; ++i) {
        const u64 ci = cs[i];
        const i64 ri = (i64)(ci >> 32);
        while (st < i && ri - (i64)(cs[st] >> 32) > kChainMaxDist) ++st;
        int max_f = kmer_size, max_j = -1, n_skip = 0;
        for (int top = i - 1; top >= st; top -= 64) {
            const int j = top - lane;
            int sc = INT32_MIN;
            bool valid = false;
            if (j >= st) {
                valid = chain_pair_score(ci, cs[j], kmer_size, f[j], &sc);
                if (valid) { const int pj = p[j]; if (pj >= 0) t[pj] = (I)i; }
                else sc = INT32_MIN;
            }
            __syncthreads();
            const bool marked = valid && t[j] == i;
            const int incl = wave_prefix_max(sc);
            int before = wave_from_below(incl, max_f);
            if (before < max_f) before = max_f;
            const bool newmax = valid && sc > before;
            const u64 NM = __ballot(newmax), SK = __ballot(marked && !newmax);
            u64 live = ~0ULL;                  // lanes the sequential loop reaches
            if (SK) {
                const u64 upto = below | (1ULL << lane);
                const int S = popc64(SK & upto) - popc64(NM & upto);
                int floor_ = wave_prefix_min(S);
                if (floor_ > -n_skip) floor_ = -n_skip;
                const int W = S - floor_;
                const u64 stop = __ballot(W > kChainMaxSkip);
                if (stop) live = (1ULL << ctz64(stop)) - 1ULL;       // the stopping lane is no new best: lanes below it count
                else n_skip = wave_read(W, 63);
            } else {
                n_skip -= popc64(NM); if (n_skip < 0) n_skip = 0;
            }
            const u64 best = NM & live;
            if (best) { const int lb = 63 - __clzll((long long)best); max_f = wave_read(sc, lb); max_j = top - lb; }
            if (live != ~0ULL) break;
        }
        if (lane == 0) {
            f[i] = (I)max_f; p[i] = (I)max_j;
            v[i] = (max_j >= 0 && v[max_j] > max_f) ? v[max_j] : (I)max_f;
        }
        __syncthreads();
    }
}

// chain_ends for one wave (n <= kLdsChain seeds).  The keys land sorted ascending in u_lds when there are at most kLdsEnds of
// them, else in u_glb; *u_out says where.  Returns the number of chain ends.
constexpr int kLdsEnds = 64;
template <class I>
NECAT_D int chain_ends_wave(const I* f, const I* p, I* t, const I* v, u64* u_lds, u64* u_glb, int n, int lane, u64** u_out)
{
    constexpr int kPer = kLdsChain / 64;
    const u64 below = (1ULL << lane) - 1ULL;
    for (int a = lane; a < n; a += 64) t[a] = 0;
    __syncthreads();
    for (int a = lane; a < n; a += 64) if (p[a] >= 0) t[p[a]] = 1;
    __syncthreads();
    u64 m[kPer]; int n_u = 0;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int a = lane + 64 * q;
        m[q] = __ballot(a < n && t[a] == 0 && v[a] >= kChainMinSc);
        n_u += popc64(m[q]);
    }
    u64* u = n_u <= kLdsEnds ? u_lds : u_glb;
    *u_out = u;
    int at = 0;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        if ((m[q] >> lane) & 1ULL) u[at + popc64(m[q] & below)] = chain_end_key(f, p, v, lane + 64 * q);     // equal keys are equal chains: their order is free
        at += popc64(m[q]);
    }
    __syncthreads();
    if (n_u > 1) {
        // rank sort in place: every lane keeps its keys in registers
        u64 key[kPer]; int rk[kPer];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int a = lane + 64 * q;
            rk[q] = -1; key[q] = 0;
            if (a < n_u) {
                const u64 ka = u[a];
                int r = 0;
                for (int b = 0; b < n_u; ++b) { const u64 kb = u[b]; r += (kb < ka) || (kb == ka && b < a); }
                rk[q] = r; key[q] = ka;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kPer; ++q) if (rk[q] >= 0) u[rk[q]] = key[q];
        __syncthreads();
    }
    return n_u;
}

#ifdef NECAT_SEED_PROF
__device__ unsigned long long g_seed_prof[32];
#define SPROF(k) do { if (lane == 0) { const u64 now_ = clock64(); pacc[k] += now_ - tprev; tprev = now_; } } while (0)
#else
#define SPROF(k) do {} while (0)
#endif

// scoring_pick (seed_core.h = word_finder.c:150-168 second half) on the lanes of a wave: only the anchor (loc[0], loc[1], its
// index) is used by the callers.  The sequential code walks [j < maxi that agree with maxi ..., maxi, j > maxi that agree ...] and
// keeps overwriting the anchor while it is "unset" - which it tests as loc == 0 - so the anchor is the first element of that
// walk with a non-zero offset, or its last element when none has one.
NECAT_D int scoring_pick_wave(const int* s_loc, const int* s_seedn, const int* s_score, int k, float scan_window, int read_size, int lane, int* msid)
{
    // k <= 2 * kBlkSeeds = 80: indices lane and lane + 64
    const int a0 = lane, a1 = lane + 64;
    const int v0 = a0 < k ? s_score[a0] : -1, v1 = a1 < k ? s_score[a1] : -1;
    const int maxval = wave_read(wave_prefix_max(v0 > v1 ? v0 : v1), 63);
    if (maxval < 5) return 0;
    const u64 e0 = __ballot(v0 == maxval), e1 = __ballot(v1 == maxval);
    const int maxi = e0 ? ctz64(e0) : 64 + ctz64(e1);
    const int rep = popc64(e0) + popc64(e1) - 1;        // later indices with the same vote
    if (rep == maxval) { *msid = maxi; return 1; }
    const int lm = s_loc[maxi], sm = s_seedn[maxi];
    auto agrees = [&](int j) {
        if (j >= k || j == maxi) return j == maxi;
        const int lj = s_loc[j], sj = s_seedn[j];
        if (j < maxi) return sm - sj > 0 && lm - lj > 0 && lm - lj < read_size && ddf_ok(lm - lj, sm - sj, scan_window);
        return sj - sm > 0 && lj - lm > 0 && lj - lm <= read_size && ddf_ok(lj - lm, sj - sm, scan_window);
    };
    const bool g0 = agrees(a0), g1 = agrees(a1);
    const u64 w0 = __ballot(g0), w1 = __ballot(g1);                                 // the walk, in index order
    const u64 n0 = __ballot(g0 && s_loc[a0 < k ? a0 : 0] != 0), n1 = __ballot(g1 && s_loc[a1 < k ? a1 : 0] != 0);
    int pick;
    if (n0) pick = ctz64(n0);
    else if (n1) pick = 64 + ctz64(n1);
    else pick = w1 ? 64 + 63 - __clzll((long long)w1) : 63 - __clzll((long long)w0);       // maxi itself is in the walk
    *msid = pick;
    return 1;
}

// seq_of_offset (dev_common.h) with 64 pivots per step instead of one
NECAT_D u64 seq_of_offset_wave(const u64* __restrict__ seq_off, u64 nseq, u64 g, int lane)
{
    u64 lo = 0, hi = nseq;       // invariant: seq_off[lo] <= g < seq_off[hi]
    while (hi - lo > 1) {
        const u64 span = hi - lo;
        // pivots lo + 1 + lane * step ... (at most 64 of them inside (lo, hi))
        const u64 step = (span + 63) / 64;
        const u64 pv = lo + 1 + (u64)lane * step;
        const bool le = pv < hi && seq_off[pv] <= g;
        const u64 m = __ballot(le);          // a prefix of the lanes (seq_off ascends)
        const int c = popc64(m);
        const u64 nlo = c ? lo + 1 + (u64)(c - 1) * step : lo;
        u64 nhi = lo + 1 + (u64)c * step; if (nhi > hi) nhi = hi;
        lo = nlo; hi = nhi;
    }
    return lo;
}

struct LdsAdder { int* s; NECAT_D void operator()(int j) { atomicAdd(&s[j], 1); } };

// Block evaluation: one WAVE per read.  The touched blocks are still visited strictly in first-touch
// order (accepted candidates zero the scores later blocks read), but inside one evaluation the O(n^2)
// DDF vote and the co-linear gather run on all 64 lanes; the order-dependent rest (anchor choice,
// chain DP, candidate choice) runs on lane 0.  __syncthreads() (the block is a single wave) separates
// the lane-0 phases from the cooperative ones - it also stops the compiler from forwarding values
// across lanes' stores.
#ifndef NECAT_SEED_WAVES
#define NECAT_SEED_WAVES 6
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NECAT_SEED_WAVES, NECAT_SEED_WAVES)))
k_seed_eval(DevVolume ref, DevVolume reads, SeedParams P, const u32* __restrict__ order, const SeedMeta* __restrict__ meta, u32 n,
            SeedArenas A, const i32* __restrict__ nblk_in, i32* __restrict__ n_strand, int* __restrict__ err_flag, int clear_ht)
{
    // LDS per wave: 6 KB, so that the CU holds as many of these latency-bound waves as their registers allow.
    // votes (phases A - C) and the gather's slot table (phase D) share one buffer
    __shared__ __attribute__((aligned(8))) int s_ab[264];
    int* const s_loc = s_ab; int* const s_seedn = s_ab + 2 * kBlkSeeds; int* const s_score = s_ab + 4 * kBlkSeeds;
    int* const s_pre = s_ab;                       // [65] seeds before slot j
    int* const s_rel = s_ab + 66;                  // [64] accepted seeds of slot j
    SBlock** const s_sb = (SBlock**)(s_ab + 130);  // [64]
    __shared__ int s_ctl[4];
    __shared__ i64 s_clear[2];
    __shared__ DevCand l_can[kLdsCan];
    // chain scratch of the common case (<= kLdsChain co-linear seeds): sort, chain DP and chain ends run on all lanes in LDS
    // (16-bit f / p / t / v: indices < 256, scores <= 256 * k)
    __shared__ u64 l_cs[kLdsChain], l_u[kLdsEnds];
    __shared__ i16 l_f[kLdsChain], l_p[kLdsChain], l_t[kLdsChain], l_v[kLdsChain];
    const u32 i = blockIdx.x >> 1;         // one wave per (read, strand): the strands share nothing but the output order
    if (i >= n) return;
    const int strand0 = (int)(blockIdx.x & 1);
    const int lane = threadIdx.x;
    const u64 below = (1ULL << lane) - 1ULL;
    const int r = (int)order[i];
    const SeedMeta m = meta[i];
    const int L = (int)(reads.seq_off[r + 1] - reads.seq_off[r]);
    const int bs = P.block_size, z = P.z, cut = P.s_cutoff;
#ifdef NECAT_SEED_PROF
    u64 pacc[10] = {0,0,0,0,0,0,0,0,0,0}; u64 tprev = clock64();
#endif
    int n_out = 0;            // meaningful on lane 0
    bool failed = false;
    for (int strand = strand0; strand == strand0; ++strand) {
        SeedScratch S = seed_scratch(A, m, strand);
        const int nblk = P.debug_phase == 1 ? 0 : nblk_in[2 * (u64)i + strand];
        // Blocks are visited in first-touch order, but only those passing the score test are evaluated and
        // a block's score only ever drops (accepted candidates zero it) while its stale score is fixed:
        // 64 blocks are tested at once, the survivors re-tested when their turn comes.
        u64 pending = 0; int pbase = -64;
        for (;;) {
            __syncthreads();
            if (!pending) {
                pbase += 64;
                if (pbase >= nblk) break;
                bool pass = false;
                if (pbase + lane < nblk) { const SBlock* q = S.pool + pbase + lane; pass = q->score >= cut && q->stale >= 2 * cut; }
                pending = __ballot(pass);
                continue;
            }
            const int bi = pbase + ctz64(pending);
            pending &= pending - 1;
            SBlock* sb = S.pool + bi;
            if (!(sb->score >= cut && sb->stale >= 2 * cut)) continue;         // wave-uniform
            SPROF(0);
            // A: seed lists (lane 0)
            // (block_seed_lists of seed_core.h: the seeds of block b - 1, if it has any, then those of block b shifted by one block)
            int ns; u64 blk_start;
            {
                const int block_id = sb->block_id;
                const SBlock* prev = sb_find(S, block_id - 1);           // the same probe on every lane
                const int np = prev ? prev->score : 0, nc = sb->score;
                if (lane < np) { s_seedn[lane] = prev->kmer_id[lane]; s_loc[lane] = prev->blk_offset[lane]; }
                if (lane < nc) { s_seedn[np + lane] = sb->kmer_id[lane]; s_loc[np + lane] = sb->blk_offset[lane] + (np ? bs : 0); }
                ns = np + nc;
                blk_start = (u64)bs * (u64)(np ? block_id - 1 : block_id);
            }
            for (int x = lane; x < kBlkSeeds * 2; x += 64) s_score[x] = 0;
            __syncthreads();
            SPROF(1);
            // B: DDF vote, one row per lane
            for (int ii = lane; ii < ns - 1; ii += 64) {
                LdsAdder add; add.s = s_score;
                const int own = scoring_vote_row(s_loc, s_seedn, ii, ns, (float)z, L, add);
                if (own) atomicAdd(&s_score[ii], own);
            }
            __syncthreads();
            SPROF(2);
            // C: anchor (lane 0)
            int msid = -1;
            if (!scoring_pick_wave(s_loc, s_seedn, s_score, ns, (float)z, L, lane, &msid)) { SPROF(3); continue; }
            if (s_score[msid] < 2 * cut) { SPROF(3); continue; }
            const i64 tid = (i64)seq_of_offset_wave(ref.seq_off, ref.nseq, (u64)s_loc[msid] + blk_start, lane);
            const AnchorGeom g = anchor_geometry(ref, s_loc[msid], s_seedn[msid], blk_start, bs, z, L, tid);     // the same on every lane
            SPROF(3);
            // D: co-linear gather (word_finder.c:249-307).  Slots = the blocks from bid_start to the anchor's own, tested on the left
            // side, then the anchor's block and the ones up to bid_end, tested on the right side.  All probes at once (one slot per
            // lane), then the seeds of all slots as ONE flat sequence taken 64 at a time: a handful of dependent memory round trips
            // per evaluation instead of four per block.  The chain seeds are sorted afterwards, so their order here is free; the
            // anchor's own seed (word_finder.c:277) goes first.
            int ncs = 1, seed_score = 0;
            bool overflow = false;
            if (lane == 0) l_cs[0] = ((u64)g.stoff << 32) | (u64)(u32)g.seed_qoff;
            {
                const int nleft = g.seed_bid - g.bid_start + 1, nslot = nleft + (g.bid_end - g.seed_bid + 1);
                for (int s0 = 0; s0 < nslot; s0 += 64) {
                    const int sl = s0 + lane;
                    const int b = sl >= nleft ? g.seed_bid + (sl - nleft) : g.bid_start + sl;
                    SBlock* sbi = sl < nslot ? sb_find(S, b) : nullptr;
                    const int nsc = sbi ? sbi->score : 0;
                    const int inc = wave_prefix_sum(nsc);
                    __syncthreads();
                    s_pre[lane] = inc - nsc; s_rel[lane] = 0; s_sb[lane] = sbi;
                    if (lane == 63) s_pre[64] = inc;
                    __syncthreads();
                    const int T = s_pre[64];
                    for (int c0 = 0; c0 < T; c0 += 64) {
                        const int q = c0 + lane;
                        int lo = 0;                                   // slot of seed q: the last one that starts at or before q
                        for (int st = 32; st > 0; st >>= 1) if (s_pre[lo + st] <= q) lo += st;
                        const int sl2 = s0 + lo;
                        const bool right = sl2 >= nleft;
                        const int b2 = right ? g.seed_bid + (sl2 - nleft) : g.bid_start + sl2;
                        u64 key = 0;
                        const bool acc = q < T && gather_test(g, s_sb[lo], q - s_pre[lo], b2, bs, z, right, &key);
                        const u64 mask = __ballot(acc);
                        if (acc) {
                            const int w = ncs + popc64(mask & below);
                            if (w < kLdsChain) l_cs[w] = key; else if ((u32)w < S.cs_cap) S.cs[w] = key;
                            atomicAdd(&s_rel[lo], 1);
                        }
                        ncs += popc64(mask);
                    }
                    __syncthreads();
                    // the 40 % rule (word_finder.c:273, :305), never for the anchor's own block
                    if (nsc && b != g.seed_bid && gather_zeroes_block(s_rel[lane], nsc)) sbi->score = 0;
                }
                seed_score = ncs - 1;
                overflow = (u32)ncs >= S.cs_cap;
            }
            __syncthreads();
            SPROF(4);
            // E: sort (all lanes when the seeds fit LDS), then chain + choose + emit
            const bool in_lds = !overflow && ncs <= kLdsChain;
            const bool wave_chain = in_lds && P.chain_wave;
            if (in_lds) {
                // rank sort in place, the keys of a lane in registers; equal keys (identical seeds) keep their index order
                constexpr int kPer = kLdsChain / 64;
                u64 key[kPer]; int rk[kPer];
#pragma unroll
                for (int q = 0; q < kPer; ++q) {
                    const int a = lane + 64 * q;
                    rk[q] = -1; key[q] = 0;
                    if (a < ncs) {
                        const u64 ka = l_cs[a];
                        int r2 = 0;
                        for (int b = 0; b < ncs; ++b) { const u64 kb = l_cs[b]; r2 += (kb < ka) || (kb == ka && b < a); }
                        rk[q] = r2; key[q] = ka;
                    }
                }
                __syncthreads();
                if (wave_chain) {
#pragma unroll
                    for (int q = 0; q < kPer; ++q) if (rk[q] >= 0) l_cs[rk[q]] = key[q];
                } else {
#pragma unroll
                    for (int q = 0; q < kPer; ++q) if (rk[q] >= 0) S.cs[rk[q]] = key[q];       // NECAT_CHAIN_WAVE=0: lane 0 chains in the global scratch
                }
            } else if (!overflow) {
                const int lim = ncs < kLdsChain ? ncs : kLdsChain;
                for (int a = lane; a < lim; a += 64) S.cs[a] = l_cs[a];
            }
            __syncthreads();
            SPROF(5);
            // chain DP and chain ends on all lanes when the seeds are in LDS
            int chained = 0; u64* u_at = l_u;
            if (wave_chain) {
                chain_fill_wave<i16>(l_cs, l_f, l_p, l_t, l_v, ncs, P.k, lane);
                chained = chain_ends_wave<i16>(l_f, l_p, l_t, l_v, l_u, S.u, ncs, lane, &u_at);
            }
            SPROF(7);
            if (lane == 0) {
                int rc;
                s_clear[0] = 1; s_clear[1] = 0;
                if (overflow) rc = kSeedErrCapacity;
                else if (wave_chain) {
                    DevCand* lc = chained <= kLdsCan ? l_can : S.lcan;
                    const int ncan = chained ? chain_emit_t<i16>(l_cs, l_f, l_p, l_t, u_at, lc, ncs, chained, P.k, P.s_cutoff, finish_proto(g, r, strand, L)) : 0;
                    rc = finish_choose(S, lc, ncan, seed_score, g, P, &n_out, s_clear);
                } else rc = finish_candidate(S, ncs, seed_score, g, P, r, strand, L, &n_out, in_lds, s_clear);
                s_ctl[2] = rc < 0 ? 1 : 0;
            }
            __syncthreads();
            // an accepted candidate zeroes the blocks it covers (word_finder.c:171-182): one probe per lane
            for (i64 cb = s_clear[0] + lane; cb <= s_clear[1]; cb += 64) { SBlock* x = sb_find(S, (i32)cb); if (x) x->score = 0; }
            SPROF(6);
#ifdef NECAT_SEED_PROF
            if (lane == 0) pacc[8] += 1;
#endif
            if (s_ctl[2]) { failed = true; break; }
        }
    }
    SPROF(9);
#ifdef NECAT_SEED_PROF

    if (lane == 0) {
        u64 tot = 0; for (int q = 0; q < 10; ++q) { atomicAdd(&g_seed_prof[q], pacc[q]); if (q != 8) tot += pacc[q]; }
        atomicMax(&g_seed_prof[10], tot); atomicMax(&g_seed_prof[11], pacc[8]);
        if (tot > 3000000) { for (int q = 0; q < 10; ++q) atomicAdd(&g_seed_prof[16 + q], pacc[q]); atomicAdd(&g_seed_prof[26], 1ULL); }     // what the long waves do
    }
#endif
    // clear_WordFindData (word_finder.c:40-52): the hash slots this strand used go back to empty, so the arena is all-empty again when the
    // call ends (what k_seed_clear did as a launch of its own, 0.18 ms per pass; nothing after this point probes the table)
    if (clear_ht) {
        __syncthreads();
        const SeedScratch S = seed_scratch(A, m, strand0);
        const int nb = nblk_in[2 * (u64)i + strand0];
        for (int b = lane; b < nb; b += 64) S.ht[S.pool[b].slot] = kHtEmpty;
    }
    if (lane == 0) {
        if (failed) atomicExch(err_flag, 1);
        n_strand[blockIdx.x] = failed ? 0 : n_out;
    }
}

// FWD candidates, then REV candidates (find_candidates is called for FWD first, pm_worker.c:100-131), then the
// per-read sort / truncation of pm_search_one_volume; one lane per read
__global__ void __launch_bounds__(64)
k_seed_finish(SeedParams P, const SeedMeta* __restrict__ meta, u32 n, SeedArenas A, const i32* __restrict__ n_strand, i32* __restrict__ n_cands)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SeedMeta m = meta[i];
    const int n0 = n_strand[2 * (u64)i], n1 = n_strand[2 * (u64)i + 1];
    DevCand* out = A.out + m.out_off;
    const DevCand* rev = out + m.out_cap0;
    for (int j = 0; j < n1; ++j) out[n0 + j] = rev[j];        // n0 <= out_cap0: ascending copy never overtakes its source
    SeedScratch S = seed_scratch(A, m, 0);
    n_cands[i] = seed_finish_read(P, S, n0 + n1);
}

// dst[final_off[i] + j] = candidate j of the i-th processed read, ids made global
__global__ void __launch_bounds__(256)
k_pack_cands(const DevCand* __restrict__ out, const SeedMeta* __restrict__ meta, const i32* __restrict__ n_cands,
             const u64* __restrict__ final_off, u32 n, int read_start_id, int ref_start_id, necat_candidate* __restrict__ dst)
{
    const u32 wave = (u32)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= n) return;
    const DevCand* src = out + meta[wave].out_off;
    necat_candidate* d = dst + final_off[wave];
    for (int j = lane; j < n_cands[wave]; j += 64) {
        const DevCand c = src[j];
        necat_candidate o;
        o.qid = c.qid + read_start_id; o.sid = c.sid + ref_start_id; o.qdir = c.qdir; o.sdir = 0; o.score = c.score; o._pad = 0;
        o.qbeg = (u64)c.qbeg; o.qend = (u64)c.qend; o.qsize = (u64)c.qsize;
        o.sbeg = (u64)c.sbeg; o.send = (u64)c.send; o.ssize = (u64)c.ssize;
        o.qoff = (u64)c.qoff; o.soff = (u64)c.soff;
        d[j] = o;
    }
}

}  // namespace necat

namespace necat {
// several seeding chunks: candidates were packed chunk by chunk (reads in work order); move every read's
// run to its place in ascending read order.  One wave per read.
__global__ void __launch_bounds__(256)
k_move_cands(const necat_candidate* __restrict__ src, const u64* __restrict__ src_off, const u64* __restrict__ dst_off,
             const i32* __restrict__ cnt, u32 n, necat_candidate* __restrict__ dst)
{
    const u32 wave = (u32)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= n) return;
    const necat_candidate* a = src + src_off[wave];
    necat_candidate* b = dst + dst_off[wave];
    for (int j = lane; j < cnt[wave]; j += 64) b[j] = a[j];
}
}  // namespace necat
