// index_kernels.h - k-mer index build on the GPU (the build_lookup_table equivalent,
// lookup_table/lookup_table.c:149).  HBM-bound integer work: count -> filter + exclusive scan ->
// scatter -> in-bucket rank.  The result arrays have exactly the reference layout
// (kmer_stats[h] = cnt<<34 | start, offset_list grouped by hash, ascending offsets inside a hash),
// so anything downstream - including the reference's own find_candidates - could consume them.
//
// Reference single-thread phases and what replaces them:
//   get_kmer_counts  (lookup_table.c:15)   -> k_count_kmers   (u32 atomics into a 4^k table)
//   cutoff + cnt<<34 (lookup_table.c:43-51)-> k_tile_sums / k_scan_partials / k_write_stats
//   get_offset_list  (lookup_table.c:60)   -> k_scatter_offsets (bucket cursor = the count itself)
//   radix_sort       (hash_list_bucket_sort.c:134, stable => ascending offsets inside a k-mer)
//                                          -> k_rank_buckets  (buckets hold <= max_occ entries)
//   build_kmer_starts/clear_hash           -> folded into k_write_stats / k_rank_buckets
#pragma once
#include <type_traits>
#include "dev_common.h"

namespace necat {

constexpr int kPosPerThread = 16;
constexpr int kScanTile = 2048;     // entries per block in the table scan (256 threads x 8)

NECAT_D u64 kmer_hash_at(const u64* bases, i64 g, int k)
{
    // 2 bits per base, earlier base more significant (lookup_table.c:8-12)
    return rev2(load32(bases, g)) >> (64 - 2 * k);
}

// the 16 consecutive positions a thread of the streaming passes hashes (g0 a multiple of 16, k <= 15: the last k-mer ends at base g0 + 29) share ONE
// 32-base window: reversed once, every position's hash is two shifts of it (the passes were bound by hashing: ~ 95 instructions per position)
NECAT_D u64 kmer_window16(const u64* bases, i64 g0) { return rev2(load32(bases, g0)); }
NECAT_D u64 kmer_hash_win(u64 win, int i, int k) { return (win << (2 * i)) >> (64 - 2 * k); }

// seq_of_offset for the threads of a block whose positions lie in [p_first, p_last]: two lanes search the whole volume once, everybody else only the
// few reads that range spans (a thread of the streaming passes used to run its own 15-step search of dependent loads for every 16 positions it hashes)
NECAT_D void block_read_range(const DevVolume& vol, u64 p_first, u64 p_last, u64* s_range)
{
    if (threadIdx.x < 2) s_range[threadIdx.x] = seq_of_offset(vol.seq_off, vol.nseq, threadIdx.x ? p_last : p_first);
    __syncthreads();
}
NECAT_D u64 seq_of_offset_in(const u64* seq_off, u64 lo, u64 hi_incl, u64 g)      // seq_off[lo] <= g < seq_off[hi_incl + 1]
{
    u64 hi = hi_incl + 1;
    while (hi - lo > 1) { const u64 mid = (lo + hi) >> 1; if (seq_off[mid] <= g) lo = mid; else hi = mid; }
    return lo;
}

// MODE 0: count occurrences.  MODE 1: scatter offsets into bucket slots; cnt32[h] then holds the
// bucket's END cursor (start + count; 0 for k-mers dropped by the occurrence cutoff), so one atomic
// both tests the k-mer and yields its slot - no second random read of kmer_stats.
template <int MODE>
__global__ void __launch_bounds__(256)
k_kmer_pass(DevVolume vol, int k, u32* __restrict__ cnt32, u64 n_offsets, u64* __restrict__ tmp_list)
{
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    const u64 nchunks = (vol.nbases + kPosPerThread - 1) / kPosPerThread;
    for (u64 ch = (u64)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunks; ch += nthreads) {
        const u64 g0 = ch * kPosPerThread;
        u64 r = seq_of_offset(vol.seq_off, vol.nseq, g0);
        u64 rend = vol.seq_off[r + 1];
#pragma unroll
        for (int i = 0; i < kPosPerThread; ++i) {
            const u64 p = g0 + i;
            if (p >= vol.nbases) break;
            while (p >= rend) { ++r; rend = vol.seq_off[r + 1]; }
            if (p + (u64)k <= rend) {       // k-mers never span reads (lookup_table.c:31-41)
                const u64 h = kmer_hash_at(vol.bases, (i64)p, k);
                if (MODE == 0) {
                    atomicAdd(&cnt32[h], 1u);
                } else {
                    const u32 old = atomicSub(&cnt32[h], 1u);      // kept k-mer: start < old <= start + count
                    if (old != 0u && (u64)old <= n_offsets) tmp_list[old - 1] = p;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Partitioned passes (k >= 11).  A k-mer's table entry is a random 4-byte cell of a 4^k table
// (4.3 GB at k = 15): 2 x N random DRAM read-modify-writes.  Instead the k-mers are first split by
// the top PB bits of their hash into NB = 2^PB buckets (a streaming pass: hash<<34|pos records written
// bucket by bucket), and the count / scatter passes then walk one bucket at a time, all blocks of a
// bucket on the same XCD (block b runs on XCD b % 8): the bucket's table slice (<= 1 MB) and its slot
// range of the offset list stay resident in that XCD's 4 MB L2 while the bucket is processed.
// ---------------------------------------------------------------------------------------------
constexpr int kPartThreads = 256;
constexpr int kPartPosPerBlock = kPartThreads * 64;   // positions of the volume one block partitions (16 and 256 per thread: slower)
constexpr int kBucketChunk = 512;                     // bucket elements one block of the bucket passes handles

// Global bucket histogram.  LDS per block: a 16-bit counter per bucket (a block sees kPartPosPerBlock = 16 384 positions, two
// counters share a word and are bumped with one 32-bit LDS atomic).
static_assert(kPartPosPerBlock < 65536, "16-bit bucket counters per block");
__global__ void __launch_bounds__(kPartThreads)
k_part_hist(DevVolume vol, int k, int shift, u32 nb, u32 b_lo, u32 b_hi, u32* __restrict__ bucket_cnt)
{
    // [b_lo, b_hi): the buckets (= the hash range) this rank builds - all of them on one GPU (necat_index_build_sharded)
    extern __shared__ u32 lds[];          // [nb / 2] packed histogram
    u32* hist = lds;
    for (u32 i = threadIdx.x; i < nb / 2; i += kPartThreads) hist[i] = 0;
    __syncthreads();
    const u64 p0 = (u64)blockIdx.x * kPartPosPerBlock;
    const u64 p1 = (p0 + kPartPosPerBlock < vol.nbases) ? p0 + kPartPosPerBlock : vol.nbases;
    __shared__ u64 s_range[2];
    block_read_range(vol, p0, p1 - 1, s_range);          // (p0 < nbases: the grid covers the volume exactly)
    const u64 r_lo = s_range[0], r_hi = s_range[1];
    for (u64 g0 = p0 + (u64)threadIdx.x * kPosPerThread; g0 < p1; g0 += (u64)kPartThreads * kPosPerThread) {
        u64 r = seq_of_offset_in(vol.seq_off, r_lo, r_hi, g0);
        u64 rend = vol.seq_off[r + 1];
        const u64 win = kmer_window16(vol.bases, (i64)g0);
#pragma unroll
        for (int i = 0; i < kPosPerThread; ++i) {
            const u64 p = g0 + i;
            if (p >= p1) break;
            while (p >= rend) { ++r; rend = vol.seq_off[r + 1]; }
            if (p + (u64)k <= rend) {
                const u32 b = (u32)(kmer_hash_win(win, i, k) >> shift);
                if (b < b_lo || b >= b_hi) continue;
                atomicAdd(&hist[b >> 1], 1u << ((b & 1u) * 16));
            }
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < nb; i += kPartThreads) { const u32 c = (hist[i >> 1] >> ((i & 1u) * 16)) & 0xffffu; if (c) atomicAdd(&bucket_cnt[i], c); }
}

// ---- the split passes.  A one-pass scatter of hash<<34|pos records into 4096 bucket regions writes 8 bytes per lane to 64
// different places per store instruction (1.6 of that pass's 2.6 ms were the stores); a split into <= 64 parts whose tile is
// first sorted by part in LDS writes runs of consecutive records from consecutive lanes instead.  So the 12 partition bits are
// split 6 + 6 (k_split_bases: volume -> coarse buckets; k_split_recs: coarse -> fine buckets), and k_subpart (the next 6 bits)
// scatters through the same staging.  The order of the records inside a part is free (k_slice_emit ranks by offset).
constexpr int kSplitTile = 4096;                  // records per tile = kPosPerThread per thread of a 256-thread block (2048: 6.2 ms for the whole build, 4096: 5.6, 8192: 5.9 -
                                                  // a tile reserves its space with up to 64 atomics on 64 cursors the whole grid shares)
constexpr int kCurStride = 16;                   // the 64 coarse cursors of k_split_bases sit on their own 128-byte lines: every block bumps all of them
static_assert(kPosPerThread == 16, "kmer_window16: 16 positions + a 15-mer = 30 bases of one 32-base window");
// T threads per tile (kSplitTile / T records each): 256 until round 5; 512 puts twice the waves behind the same 33 KB of LDS - the split kernels are chains of
// short barrier-separated phases (count, scan, stage, copy out) and four workgroups of four waves per CU did not hide their latencies
struct SplitLds { u64 rec[kSplitTile]; u64 gbase[64]; u32 cnt[64]; u32 lstart[65]; };

// One tile: r[q] (valid if bit q of `valid`) -> part digit(rec) in [0, 64); reserve(d, c) = where the tile's c records of part
// d go in `out` (called by lane d of wave 0 for the parts with c > 0).  All 256 threads call.
template <int T, class Digit, class Reserve>
NECAT_D void split_tile(SplitLds& L, const u64 (&r)[kSplitTile / T], u32 valid, Digit digit, Reserve reserve, u64* __restrict__ out)
{
    constexpr int kSplitPer = kSplitTile / T;
    const int tid = threadIdx.x;
    if (tid < 64) L.cnt[tid] = 0;
    __syncthreads();
    u32 rk[kSplitPer];
#pragma unroll
    for (int q = 0; q < kSplitPer; ++q) rk[q] = (valid >> q) & 1u ? atomicAdd(&L.cnt[digit(r[q])], 1u) : 0u;
    __syncthreads();
    if (tid < 64) {
        const u32 c = L.cnt[tid];
        const u32 incl = wave_scan_add(c);
        L.lstart[tid] = incl - c;
        if (tid == 63) L.lstart[64] = incl;
        L.gbase[tid] = c ? reserve((u32)tid, c) : 0ULL;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kSplitPer; ++q) if ((valid >> q) & 1u) L.rec[L.lstart[digit(r[q])] + rk[q]] = r[q];
    __syncthreads();
    const u32 total = L.lstart[64];
    for (u32 idx = tid; idx < total; idx += T) {
        const u64 rec = L.rec[idx];
        const u32 d = digit(rec);
        out[L.gbase[d] + (idx - L.lstart[d])] = rec;
    }
    __syncthreads();
}

// volume -> records in coarse buckets (top bits1 of the bucket id); cursor[c] runs from the coarse bucket's start
template <int T>
__global__ void __launch_bounds__(T)
k_split_bases(DevVolume vol, int k, int shift, u32 b_lo, u32 b_hi, int bits2, u64* __restrict__ cursor, int cur_stride, u64* __restrict__ out)
{
    constexpr int kSplitPer = kSplitTile / T;
    static_assert(kSplitPer <= kPosPerThread, "kmer_window16 covers a thread's positions");
    __shared__ SplitLds L;
    __shared__ u64 s_range[2];
    const u64 t0 = (u64)blockIdx.x * kSplitTile, t1 = t0 + kSplitTile < vol.nbases ? t0 + kSplitTile : vol.nbases;
    block_read_range(vol, t0, t1 - 1, s_range);
    const u64 g0 = t0 + (u64)threadIdx.x * kSplitPer;
    u64 r[kSplitPer]; u32 valid = 0;
    if (g0 < vol.nbases) {
        u64 sq = seq_of_offset_in(vol.seq_off, s_range[0], s_range[1], g0);
        u64 rend = vol.seq_off[sq + 1];
        const u64 win = kmer_window16(vol.bases, (i64)g0);
#pragma unroll
        for (int i = 0; i < kSplitPer; ++i) {
            const u64 p = g0 + i;
            r[i] = 0;
            if (p >= vol.nbases) continue;
            while (p >= rend) { ++sq; rend = vol.seq_off[sq + 1]; }
            if (p + (u64)k <= rend) {
                const u64 h = kmer_hash_win(win, i, k);
                const u32 b = (u32)(h >> shift);
                if (b >= b_lo && b < b_hi) { r[i] = (h << kOffsetBits) | p; valid |= 1u << i; }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < kSplitPer; ++i) r[i] = 0;
    }
    const int dsh = kOffsetBits + shift + bits2;
    split_tile<T>(L, r, valid, [=](u64 rec) { return (u32)(rec >> dsh) & 63u; },
               [=](u32 d, u32 c) { return (u64)atomicAdd(reinterpret_cast<unsigned long long*>(&cursor[(u64)d * cur_stride]), (unsigned long long)c); }, out);
}

// tiles of the coarse buckets: tile_pre[c] = tiles of the coarse buckets before c (k_bucket_scan), cstart = bucket_start
// at stride 1 << bits2.  Records of coarse bucket c -> its fine buckets (c << bits2) + d through bucket_cursor.
template <int T>
__global__ void __launch_bounds__(T)
k_split_recs(const u64* __restrict__ in, const u64* __restrict__ bucket_start, const u32* __restrict__ tile_pre, int nc, int shift, int bits2,
             u64* __restrict__ bucket_cursor, u64* __restrict__ out)
{
    constexpr int kSplitPer = kSplitTile / T;
    __shared__ SplitLds L;
    const u32 t = blockIdx.x;
    if (t >= tile_pre[nc]) return;
    int c = 0;
    for (int st = 32; st > 0; st >>= 1) if (c + st < nc && tile_pre[c + st] <= t) c += st;       // the last c with tile_pre[c] <= t
    const u64 lo = bucket_start[(u64)c << bits2] + (u64)(t - tile_pre[c]) * kSplitTile, hi = bucket_start[(u64)(c + 1) << bits2];
    u64 r[kSplitPer]; u32 valid = 0;
#pragma unroll
    for (int q = 0; q < kSplitPer; ++q) { const u64 e = lo + (u64)q * T + threadIdx.x; r[q] = 0; if (e < hi) { r[q] = in[e]; valid |= 1u << q; } }
    const int dsh = kOffsetBits + shift;
    const u32 dmask = (1u << bits2) - 1u;
    u64* cur = bucket_cursor + ((u64)c << bits2);
    split_tile<T>(L, r, valid, [=](u64 rec) { return (u32)(rec >> dsh) & dmask; },
               [=](u32 d, u32 n) { return (u64)atomicAdd(reinterpret_cast<unsigned long long*>(&cur[d]), (unsigned long long)n); }, out);
}

// exclusive scan of the bucket histogram (nb <= 4096) -> bucket_start[nb + 1]; cursors start there.  For the split passes:
// coarse_cur[c] = start of coarse bucket c (= bucket_start[c << bits2]), tile_pre[c] = kSplitTile-record tiles of the coarse
// buckets before c (tile_pre[nc] = all of them).
__global__ void __launch_bounds__(1024)
k_bucket_scan(const u32* __restrict__ bucket_cnt, u32 nb, u64* __restrict__ bucket_start, u64* __restrict__ bucket_cursor,
              int bits2, u64* __restrict__ coarse_cur, u32* __restrict__ tile_pre)
{
    __shared__ u64 sh[1024];
    const u32 per = (nb + 1023) / 1024;
    const u32 lo = threadIdx.x * per, hi = (lo + per < nb) ? lo + per : nb;
    u64 s = 0;
    for (u32 i = lo; i < hi; ++i) s += bucket_cnt[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { u64 run = 0; for (int i = 0; i < 1024; ++i) { const u64 v = sh[i]; sh[i] = run; run += v; } bucket_start[nb] = run; }
    __syncthreads();
    u64 run = sh[threadIdx.x];
    for (u32 i = lo; i < hi; ++i) { bucket_start[i] = run; bucket_cursor[i] = run; run += bucket_cnt[i]; }
    __syncthreads();
    if (coarse_cur && threadIdx.x == 0) {
        const u32 nc = nb >> bits2;
        u32 tiles = 0;
        for (u32 c = 0; c < nc; ++c) {
            const u64 a = bucket_start[(u64)c << bits2], b = bucket_start[(u64)(c + 1) << bits2];
            coarse_cur[(u64)c * kCurStride] = a; tile_pre[c] = tiles;
            tiles += (u32)((b - a + kSplitTile - 1) / kSplitTile);
        }
        tile_pre[nc] = tiles;
    }
}

// MODE 0: count (atomicAdd on the table).  MODE 1: scatter offsets through the end cursors.
// Block b -> XCD x = b % 8, sequence number g = b / 8 on that XCD -> bucket (g / chunks) * 8 + x,
// chunk g % chunks: consecutive blocks of one XCD work on the same bucket.
template <int MODE>
__global__ void __launch_bounds__(256)
k_bucket_pass(const u64* __restrict__ part, const u64* __restrict__ bucket_start, u32 nb, u32 chunks,
              u32* __restrict__ cnt32, u64 n_offsets, u64* __restrict__ tmp_list)
{
    const u32 x = blockIdx.x & 7u, g = blockIdx.x >> 3;
    const u32 bucket = (g / chunks) * 8u + x;
    if (bucket >= nb) return;
    const u64 lo = bucket_start[bucket], hi = bucket_start[bucket + 1];
    for (u64 c = g % chunks; lo + c * kBucketChunk < hi; c += chunks) {
        const u64 e0 = lo + c * kBucketChunk;
        for (u64 e = e0 + threadIdx.x; e < e0 + kBucketChunk && e < hi; e += 256) {
            const u64 rec = part[e];
            const u64 h = rec >> kOffsetBits;
            if (MODE == 0) atomicAdd(&cnt32[h], 1u);
            else {
                const u32 old = atomicSub(&cnt32[h], 1u);
                if (old != 0u && (u64)old <= n_offsets) tmp_list[old - 1] = rec & kOffsetMask;
            }
        }
    }
}

NECAT_D u32 filtered_count(u32 c, u32 max_occ) { return c > max_occ ? 0u : c; }   // lookup_table.c:44

__global__ void __launch_bounds__(256)
k_tile_sums(const u32* __restrict__ cnt32, u64 n, u32 max_occ, u64* __restrict__ partial)
{
    __shared__ u64 red[256];
    const u64 base = (u64)blockIdx.x * kScanTile;
    u64 s = 0;
    for (int i = threadIdx.x; i < kScanTile; i += 256) {
        const u64 idx = base + i;
        if (idx < n) s += filtered_count(cnt32[idx], max_occ);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// exclusive scan of the tile sums, in place; total to partial[n]
__global__ void __launch_bounds__(1024)
k_scan_partials(u64* __restrict__ partial, u64 n)
{
    __shared__ u64 sh[1024];
    const u64 per = (n + 1023) / 1024;
    const u64 lo = (u64)threadIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    u64 s = 0;
    for (u64 i = lo; i < hi; ++i) s += partial[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { u64 run = 0; for (int i = 0; i < 1024; ++i) { u64 v = sh[i]; sh[i] = run; run += v; } partial[n] = run; }
    __syncthreads();
    u64 run = sh[threadIdx.x];
    for (u64 i = lo; i < hi; ++i) { u64 v = partial[i]; partial[i] = run; run += v; }
}

// kmer_stats[h] = cnt<<34 | start (start = 0 for absent k-mers, as in the reference where only
// present hashes get their start OR-ed in: lookup_table.c:94-113); cnt32[h] becomes the scatter
// pass's end cursor.  Pure streaming: 4 B in, 12 B out per table entry.
__global__ void __launch_bounds__(256)
k_write_stats(u32* __restrict__ cnt32, u64 n, u32 max_occ, const u64* __restrict__ partial, u64* __restrict__ kmer_stats)
{
    __shared__ u64 wave_tot[4];
    const u64 base = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 c[8]; u64 s = 0;
    if (base + 8 <= n) {
        const uint4 v0 = *reinterpret_cast<const uint4*>(cnt32 + base), v1 = *reinterpret_cast<const uint4*>(cnt32 + base + 4);
        c[0] = v0.x; c[1] = v0.y; c[2] = v0.z; c[3] = v0.w; c[4] = v1.x; c[5] = v1.y; c[6] = v1.z; c[7] = v1.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = base + i < n ? cnt32[base + i] : 0u;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { c[i] = filtered_count(c[i], max_occ); s += c[i]; }
    u64 incl = s;                                  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u64 v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    u64 run = partial[blockIdx.x] + incl - s;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
    u64 st[8]; u32 cur[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        st[i] = ((u64)c[i] << kOffsetBits) | (c[i] ? run : 0ULL);
        run += c[i];
        cur[i] = c[i] ? (u32)run : 0u;             // end cursor = start + count
    }
    if (base + 8 <= n) {
        ulonglong2* o = reinterpret_cast<ulonglong2*>(kmer_stats + base);
        o[0] = make_ulonglong2(st[0], st[1]); o[1] = make_ulonglong2(st[2], st[3]);
        o[2] = make_ulonglong2(st[4], st[5]); o[3] = make_ulonglong2(st[6], st[7]);
        uint4* q = reinterpret_cast<uint4*>(cnt32 + base);
        q[0] = make_uint4(cur[0], cur[1], cur[2], cur[3]); q[1] = make_uint4(cur[4], cur[5], cur[6], cur[7]);
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) if (base + i < n) { kmer_stats[base + i] = st[i]; cnt32[base + i] = cur[i]; }
    }
}

// offset_list[start + rank] = offset, rank = number of smaller offsets in the same bucket
__global__ void __launch_bounds__(256)
k_rank_buckets(DevVolume vol, int k, const u64* __restrict__ kmer_stats, const u64* __restrict__ tmp_list, u64 n, u64* __restrict__ offset_list)
{
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthreads) {
        const u64 p = tmp_list[i];
        const u64 st = kmer_stats[kmer_hash_at(vol.bases, (i64)p, k)];
        const u64 start = st & kOffsetMask, cnt = st >> kOffsetBits;
        u64 rank = 0;
        if (cnt > 1) { for (u64 j = 0; j < cnt; ++j) rank += tmp_list[start + j] < p; }
        offset_list[start + rank] = p;
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-slice passes (the default for k >= 11).  After the bucket partition (2^18 table entries per
// bucket) the records of every bucket are split once more by the next 6 hash bits; a sub-bucket then
// covers a SLICE of 4096 consecutive table entries whose counters fit LDS (16 KB).  One workgroup per
// slice counts its records with LDS atomics, scans the kept counts, writes its 32 KB of kmer_stats in
// one coalesced sweep (the table is written exactly once, never zeroed, never re-read), hands every kept
// record a slot through an LDS cursor and finally ranks the records of multi-occurrence k-mers by
// offset.  No global atomic and no dense pass over the 4^k table is left.
// ---------------------------------------------------------------------------------------------
constexpr int kSubBits = 6, kSubs = 1 << kSubBits;          // sub-buckets per bucket
constexpr int kSliceBits = 12, kSlice = 1 << kSliceBits;    // table entries per slice
static_assert(kSubBits + kSliceBits == 18, "a bucket holds 2^18 table entries");

// part -> part2: the records of bucket b regrouped by sub-bucket; sub_start[b * 64 + j] = first record
// of slice (b, j) in part2 (absolute), sub_start[nb * 64] = number of records.  One workgroup per bucket: a histogram pass,
// then the scatter tile by tile through LDS (split_tile) with running cursors.
template <int T>
__global__ void __launch_bounds__(T)
k_subpart(const u64* __restrict__ part, const u64* __restrict__ bucket_start, u32 nb, u64* __restrict__ part2, u64* __restrict__ sub_start)
{
    constexpr int kSplitPer = kSplitTile / T;
    __shared__ SplitLds L;
    __shared__ u32 hist[T / 64][kSubs];
    __shared__ u64 run[kSubs];
    const u32 b = blockIdx.x;
    const u64 lo = bucket_start[b], hi = bucket_start[b + 1];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    hist[w][lane] = 0;
    __syncthreads();
    constexpr int kSubUnroll = 4;
    for (u64 e0 = lo + (u64)w * 64; e0 < hi; e0 += T * kSubUnroll) {
        u64 rec[kSubUnroll];
#pragma unroll
        for (int q = 0; q < kSubUnroll; ++q) { const u64 e = e0 + T * q + lane; rec[q] = e < hi ? part[e] : ~0ULL; }
#pragma unroll
        for (int q = 0; q < kSubUnroll; ++q) if (e0 + T * q + lane < hi) atomicAdd(&hist[w][(u32)(rec[q] >> (kOffsetBits + kSliceBits)) & (kSubs - 1)], 1u);
    }
    __syncthreads();
    if (w == 0) {
        u32 tot = 0;
#pragma unroll
        for (int ww = 0; ww < T / 64; ++ww) tot += hist[ww][lane];
        const u32 incl = wave_scan_add(tot);
        run[lane] = lo + (incl - tot);
        sub_start[(u64)b * kSubs + lane] = lo + (incl - tot);
        if (b == nb - 1 && lane == 0) sub_start[(u64)nb * kSubs] = hi;
    }
    __syncthreads();
    for (u64 t0 = lo; t0 < hi; t0 += kSplitTile) {
        u64 r[kSplitPer]; u32 valid = 0;
#pragma unroll
        for (int q = 0; q < kSplitPer; ++q) { const u64 e = t0 + (u64)q * T + threadIdx.x; r[q] = 0; if (e < hi) { r[q] = part[e]; valid |= 1u << q; } }
        split_tile<T>(L, r, valid, [](u64 rec) { return (u32)(rec >> (kOffsetBits + kSliceBits)) & (u32)(kSubs - 1); },
                   [&](u32 d, u32 c) { const u64 at = run[d]; run[d] = at + c; return at; }, part2);
    }
}

template <int T = 256>
NECAT_D void slice_count(const u64* __restrict__ part2, u64 lo, u64 hi, u32* cnt)
{
    {
        uint4* z = reinterpret_cast<uint4*>(cnt) + threadIdx.x * (kSlice / T / 4);        // (cnt: 16-byte aligned)
#pragma unroll
        for (int i = 0; i < kSlice / T / 4; ++i) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    for (u64 e = lo + threadIdx.x; e < hi; e += T) atomicAdd(&cnt[(u32)(part2[e] >> kOffsetBits) & (kSlice - 1)], 1u);
    __syncthreads();
}

// kept_tot[s] = number of offset-list entries slice s contributes (k-mers with 1..max_occ occurrences), pres_tot[s] = number of
// those k-mers = entries of the compact table (+ both summed per bucket, so that the scans that follow run over <= 4096 values,
// not 262 144)
__global__ void __launch_bounds__(256)
k_slice_count(const u64* __restrict__ part2, const u64* __restrict__ sub_start, u32 max_occ, u32* __restrict__ kept_tot, u32* __restrict__ bucket_kept,
              u32* __restrict__ pres_tot, u32* __restrict__ bucket_pres, u32 s0)
{
    __shared__ __attribute__((aligned(16))) u32 cnt[kSlice];
    __shared__ u32 red[4], redp[4];
    const u64 s = (u64)blockIdx.x + s0;           // s0 = first slice of this rank's hash range
    slice_count(part2, sub_start[s], sub_start[s + 1], cnt);
    u32 sum = 0, pres = 0;
    {
        const uint4* own = reinterpret_cast<const uint4*>(cnt) + threadIdx.x * (kSlice / 256 / 4);      // any 16 entries: only the sums matter
#pragma unroll
        for (int i = 0; i < kSlice / 256 / 4; ++i) {
            const uint4 v = own[i];
            const u32 c0 = filtered_count(v.x, max_occ), c1 = filtered_count(v.y, max_occ), c2 = filtered_count(v.z, max_occ), c3 = filtered_count(v.w, max_occ);
            sum += c0 + c1 + c2 + c3; pres += (c0 ? 1u : 0u) + (c1 ? 1u : 0u) + (c2 ? 1u : 0u) + (c3 ? 1u : 0u);
        }
    }
    sum = wave_last(wave_scan_add(sum)); pres = wave_last(wave_scan_add(pres));
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = sum; redp[threadIdx.x >> 6] = pres; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 t = red[0] + red[1] + red[2] + red[3], q = redp[0] + redp[1] + redp[2] + redp[3];
        kept_tot[s] = t; pres_tot[s] = q;
        if (t) { atomicAdd(&bucket_kept[s >> kSubBits], t); atomicAdd(&bucket_pres[s >> kSubBits], q); }
    }
}

// exclusive scan of bucket_kept[nb] (nb <= 4096) -> bucket_base[nb + 1] (u64)
__global__ void __launch_bounds__(1024)
k_bucket_base(const u32* __restrict__ bucket_kept, u32 nb, u64* __restrict__ bucket_base)
{
    __shared__ u64 sh[1024];
    const u32 per = (nb + 1023) / 1024;
    const u32 lo = threadIdx.x * per, hi = (lo + per < nb) ? lo + per : nb;
    u64 s = 0;
    for (u32 i = lo; i < hi; ++i) s += bucket_kept[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { const u64 v = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0ULL; __syncthreads(); sh[threadIdx.x] += v; __syncthreads(); }
    u64 run = sh[threadIdx.x] - s;
    for (u32 i = lo; i < hi; ++i) { bucket_base[i] = run; run += bucket_kept[i]; }
    if (threadIdx.x == 1023) bucket_base[nb] = sh[1023];
}

// The slice's part of the sparse table (IdxWord per 64 entries + the non-zero entries, dev_common.h) and of the offset list (tmp:
// same layout, order inside a k-mer not yet fixed).
// The kernel is a chain of short barrier-separated phases, latency bound at full occupancy, so the common case - a slice of at
// most 2 T records of which at most kLdsTmp are kept - keeps its records in registers from the first load on and ranks inside
// LDS; anything bigger reloads per phase and ranks through the global tmp array.
// LT: offsets of a slice that are ranked inside LDS.  2048 (with the register path above) at E. coli size, 700 records per slice; a volume at
// oc2mkdb's 2 Gbp cut has 7 600 per slice, every slice took the reload + global tmp path and this kernel was 50 of the build's 90 ms:
// with LT = 8000 (64 KB of LDS per workgroup in all) such a slice still reads its records three times - coalesced - but groups and ranks
// them in LDS.
constexpr int kLdsTmp = 2016, kLdsTmpBig = 8000;      // (2016, not 2048: with cnt + cur + the small arrays that is 40 912 bytes, and FOUR workgroups fit a CU's 160 KB - 2048 was 80 bytes too many for the fourth)
template <int T, int LT = kLdsTmp>
__global__ void __launch_bounds__(T)
k_slice_emit(const u64* __restrict__ part2, const u64* __restrict__ sub_start, u32 max_occ, const u64* __restrict__ bucket_base, const u32* __restrict__ kept_tot,
             const u64* __restrict__ bucket_cbase, const u32* __restrict__ pres_tot, IdxWord* __restrict__ words, u64* __restrict__ compact,
             u32* __restrict__ tmp, u64* __restrict__ offset_list, u32 s0, u64 base_add, u64 cbase_add)
{
    // s0 / base_add / cbase_add: first slice of this rank's hash range / offset-list entries / compact entries of the ranks before it
    // (the starts and bases written here are final: positions in the gathered arrays; tmp and compact are addressed the same way by
    // pointers shifted back by base_add / cbase_add)
    static_assert(kSlice / T == 8, "a thread owns the 8 table entries of one byte of an IdxWord");
    __shared__ __attribute__((aligned(16))) u32 cnt[kSlice];      // occurrences per table entry of the slice
    __shared__ __attribute__((aligned(16))) u32 cur[kSlice];      // start of the entry's group inside the slice, then its fill cursor
    __shared__ u32 ltmp[LT];         // the kept offsets of the slice, grouped by entry (small slices)
    __shared__ u32 wtot[T / 64], ptot[T / 64];
    __shared__ u64 s_base, s_cbase;
    const u64 s = (u64)blockIdx.x + s0;
    const u64 lo = sub_start[s], hi = sub_start[s + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool small = hi - lo <= (u64)(2 * T);
    // the records first: their loads fly while the counters are zeroed and the slice's bases are summed
    u64 r0 = ~0ULL, r1 = ~0ULL;
    if (small) {
        if (lo + threadIdx.x < hi) r0 = part2[lo + threadIdx.x];
        if (lo + T + threadIdx.x < hi) r1 = part2[lo + T + threadIdx.x];
    }
    if (wave == T / 64 - 1) {        // the slice's bases = its bucket's bases + the totals of the bucket's earlier slices
        const u32 j = (u32)s & (kSubs - 1);
        const u64 first = s & ~(u64)(kSubs - 1);
        // (a bucket's kept entries are < 2^32: offsets are 32-bit positions of one volume)
        const u32 before = wave_last(wave_scan_add((u32)lane < j ? kept_tot[first + lane] : 0u)), pbefore = wave_last(wave_scan_add((u32)lane < j ? pres_tot[first + lane] : 0u));
        if (lane == 0) { s_base = bucket_base[s >> kSubBits] + before + base_add; s_cbase = bucket_cbase[s >> kSubBits] + pbefore + cbase_add; }
    }
    {
        uint4* z = reinterpret_cast<uint4*>(cnt) + threadIdx.x * (kSlice / T / 4);
#pragma unroll
        for (int i = 0; i < kSlice / T / 4; ++i) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    if (small) {
        if (r0 != ~0ULL) atomicAdd(&cnt[(u32)(r0 >> kOffsetBits) & (kSlice - 1)], 1u);
        if (r1 != ~0ULL) atomicAdd(&cnt[(u32)(r1 >> kOffsetBits) & (kSlice - 1)], 1u);
    } else {
        for (u64 e = lo + threadIdx.x; e < hi; e += T) atomicAdd(&cnt[(u32)(part2[e] >> kOffsetBits) & (kSlice - 1)], 1u);
    }
    __syncthreads();
    const u64 base = s_base, cbase = s_cbase;
    // exclusive scans of the kept counts and of the non-zero entries: thread t owns entries [8 t, 8 t + 8)
    constexpr int E = kSlice / T;
    u32 c[E], sum = 0, pres = 0, fb = 0;
    {
        const uint4* own = reinterpret_cast<const uint4*>(cnt) + threadIdx.x * (E / 4);
        const uint4 v0 = own[0], v1 = own[1];
        c[0] = v0.x; c[1] = v0.y; c[2] = v0.z; c[3] = v0.w; c[4] = v1.x; c[5] = v1.y; c[6] = v1.z; c[7] = v1.w;
    }
#pragma unroll
    for (int i = 0; i < E; ++i) { c[i] = filtered_count(c[i], max_occ); sum += c[i]; if (c[i]) { ++pres; fb |= 1u << i; } }
    const u32 incl = wave_scan_add(sum), pincl = wave_scan_add(pres);
    if (lane == 63) { wtot[wave] = incl; ptot[wave] = pincl; }
    __syncthreads();
    u32 run = incl - sum, prun = pincl - pres, kept_all = 0;
#pragma unroll
    for (int w = 0; w < T / 64; ++w) { if (w < wave) { run += wtot[w]; prun += ptot[w]; } kept_all += wtot[w]; }
    // kmer_stats[h] = cnt<<34 | start for the k-mers that exist and pass the cutoff (lookup_table.c:43-51, :94-113): the non-zero
    // entries in hash order; the word of 64 entries = the flag bytes of 8 neighbouring threads
    {
        const u32 sh8 = 8u * ((u32)lane & 3u);
        const u32 bl = or_lanes8((lane & 4) ? 0u : fb << sh8), bh = or_lanes8((lane & 4) ? fb << sh8 : 0u);      // lanes 0 - 3 of the 8: bytes 0 - 3, lanes 4 - 7: bytes 4 - 7
        if ((lane & 7) == 0) { IdxWord w; w.bits = ((u64)bh << 32) | bl; w.base = cbase + prun; words[s * (kSlice / 64) + (threadIdx.x >> 3)] = w; }
        u32 a[E];
        { u32 at = run;
#pragma unroll
          for (int i = 0; i < E; ++i) { a[i] = at; at += c[i]; } }
        uint4* own = reinterpret_cast<uint4*>(cur) + threadIdx.x * (E / 4);
        own[0] = make_uint4(a[0], a[1], a[2], a[3]); own[1] = make_uint4(a[4], a[5], a[6], a[7]);
        u64 q = cbase + prun;
#pragma unroll
        for (int i = 0; i < E; ++i) if (c[i]) compact[q++] = ((u64)c[i] << kOffsetBits) | (base + a[i]);
    }
    __syncthreads();
    if (small && kept_all <= (u32)LT) {
        // radix_sort is stable (hash_list_bucket_sort.c:134): offsets ascend inside a k-mer
        const u32 h0 = (u32)(r0 >> kOffsetBits) & (kSlice - 1), h1 = (u32)(r1 >> kOffsetBits) & (kSlice - 1);
        const u32 k0 = r0 != ~0ULL ? filtered_count(cnt[h0], max_occ) : 0u, k1 = r1 != ~0ULL ? filtered_count(cnt[h1], max_occ) : 0u;
        const u32 p0 = (u32)(r0 & kOffsetMask), p1 = (u32)(r1 & kOffsetMask);
        if (k0) ltmp[atomicAdd(&cur[h0], 1u)] = p0;
        if (k1) ltmp[atomicAdd(&cur[h1], 1u)] = p1;
        __syncthreads();        // cur[h] is now the END of the group
        if (k0) { const u32 st = cur[h0] - k0; u32 rank = 0; if (k0 > 1) for (u32 j = 0; j < k0; ++j) rank += ltmp[st + j] < p0; offset_list[base + st + rank] = (u64)p0; }
        if (k1) { const u32 st = cur[h1] - k1; u32 rank = 0; if (k1 > 1) for (u32 j = 0; j < k1; ++j) rank += ltmp[st + j] < p1; offset_list[base + st + rank] = (u64)p1; }
        return;
    }
    if (kept_all <= (u32)LT) {
        // a bigger slice whose kept offsets still fit LDS: the records come from global memory again, the groups and the ranks stay in LDS
        for (u64 e = lo + threadIdx.x; e < hi; e += T) {
            const u64 rec = part2[e];
            const u32 h = (u32)(rec >> kOffsetBits) & (kSlice - 1);
            if (filtered_count(cnt[h], max_occ)) ltmp[atomicAdd(&cur[h], 1u)] = (u32)(rec & kOffsetMask);
        }
        __syncthreads();
        for (u64 e = lo + threadIdx.x; e < hi; e += T) {
            const u64 rec = part2[e];
            const u32 h = (u32)(rec >> kOffsetBits) & (kSlice - 1);
            const u32 k = filtered_count(cnt[h], max_occ);
            if (!k) continue;
            const u32 p = (u32)(rec & kOffsetMask);
            const u32 st = cur[h] - k;
            u32 rank = 0;
            if (k > 1) for (u32 j = 0; j < k; ++j) rank += ltmp[st + j] < p;
            offset_list[base + st + rank] = (u64)p;
        }
        return;
    }
    for (u64 e = lo + threadIdx.x; e < hi; e += T) {
        const u64 rec = part2[e];
        const u32 h = (u32)(rec >> kOffsetBits) & (kSlice - 1);
        if (filtered_count(cnt[h], max_occ)) tmp[base + atomicAdd(&cur[h], 1u)] = (u32)(rec & kOffsetMask);
    }
    __syncthreads();        // cur[h] is now the END of the group; the tmp writes of this workgroup are visible to it
    for (u64 e = lo + threadIdx.x; e < hi; e += T) {
        const u64 rec = part2[e];
        const u32 h = (u32)(rec >> kOffsetBits) & (kSlice - 1);
        const u32 k = filtered_count(cnt[h], max_occ);
        if (!k) continue;
        const u32 p = (u32)(rec & kOffsetMask);
        const u64 st = base + cur[h] - k;
        u32 rank = 0;
        if (k > 1) for (u32 j = 0; j < k; ++j) rank += tmp[st + j] < p;
        offset_list[st + rank] = (u64)p;
    }
}

// the sparse table written out in the reference layout (necat_index_download): dense[h] = kmer_stats[h]
__global__ void __launch_bounds__(256)
k_index_expand(IndexView index, u64 n, u64* __restrict__ dense)
{
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    for (u64 h = (u64)blockIdx.x * blockDim.x + threadIdx.x; h < n; h += nthreads) dense[h] = index.lookup(h);
}

// ASCII bases -> NECAT pac bytes (pdb_add_one_seq / _set_pac, packed_db.c:229-252, with common/nst_nt4_table.c: A C G T in either
// case -> 0..3, '-' -> 5, everything else -> 4): the code is OR-ed in at the base's 2-bit slot, first base of a byte in its top
// bits; codes 4 and 5 spill into the neighbouring slot (or out of the byte) exactly as in the reference.  One thread per pac byte.
NECAT_D u32 nt4_code(u32 ch)
{
    const u32 u = ch & 0xdfu;                  // upper case for letters
    if (u == 'A') return 0u;
    if (u == 'C') return 1u;
    if (u == 'G') return 2u;
    if (u == 'T') return 3u;
    return ch == '-' ? 5u : 4u;
}
__global__ void __launch_bounds__(256)
k_pack_ascii(const unsigned char* __restrict__ ascii, u64 nbases, u64 base0, unsigned char* __restrict__ pac)
{
    // base0 (a multiple of 4): the first base of this piece within the volume; ascii / pac point at the piece
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    const u64 nbytes = (nbases + 3) / 4;
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < nbytes; j += nthreads) {
        u32 b = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const u64 i = 4 * j + q; if (i < nbases) b |= nt4_code(ascii[i]) << (6 - 2 * q); }
        pac[j] = (unsigned char)b;
    }
    (void)base0;
}

// NECAT pac (first base of a byte in its top two bits) -> little-endian 2-bit words
__global__ void __launch_bounds__(256)
k_repack(const u64* __restrict__ pac_words, u64 nwords, u64* __restrict__ out)
{
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += nthreads) {
        const u64 w = pac_words[i];   // byte j of the word = pac byte 8*i + j (little-endian load)
        out[i] = ((w & 0x0303030303030303ULL) << 6) | ((w & 0x0C0C0C0C0C0C0C0CULL) << 2) |
                 ((w >> 2) & 0x0C0C0C0C0C0C0C0CULL) | ((w >> 6) & 0x0303030303030303ULL);
    }
}

}  // namespace necat
