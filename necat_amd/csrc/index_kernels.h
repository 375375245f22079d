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
#include "dev_common.h"

namespace necat {

constexpr int kPosPerThread = 8;
constexpr int kScanTile = 2048;     // entries per block in the table scan (256 threads x 8)

NECAT_D u64 kmer_hash_at(const u64* bases, i64 g, int k)
{
    // 2 bits per base, earlier base more significant (lookup_table.c:8-12)
    return rev2(load32(bases, g)) >> (64 - 2 * k);
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
constexpr int kPartPosPerBlock = kPartThreads * 64;   // positions of the volume one block partitions
constexpr int kBucketChunk = 512;                     // bucket elements one block of the bucket passes handles

// MODE 0: global bucket histogram.  MODE 1: scatter hash<<34|pos records into the bucket regions.
template <int MODE>
__global__ void __launch_bounds__(kPartThreads)
k_part_pass(DevVolume vol, int k, int shift, u32 nb, u32 b_lo, u32 b_hi, u32* __restrict__ bucket_cnt, u64* __restrict__ bucket_cursor, u64* __restrict__ part)
{
    // [b_lo, b_hi): the buckets (= the hash range) this rank builds - all of them on one GPU (necat_index_build_sharded)
    extern __shared__ u32 lds[];          // [nb] histogram (+ [2*nb] 64-bit bases in MODE 1)
    u32* hist = lds;
    u64* base = reinterpret_cast<u64*>(lds + nb);
    for (u32 i = threadIdx.x; i < nb; i += kPartThreads) hist[i] = 0;
    __syncthreads();
    const u64 p0 = (u64)blockIdx.x * kPartPosPerBlock;
    const u64 p1 = (p0 + kPartPosPerBlock < vol.nbases) ? p0 + kPartPosPerBlock : vol.nbases;
    for (int pass = 0; pass < (MODE == 0 ? 1 : 2); ++pass) {
        for (u64 g0 = p0 + (u64)threadIdx.x * kPosPerThread; g0 < p1; g0 += (u64)kPartThreads * kPosPerThread) {
            u64 r = seq_of_offset(vol.seq_off, vol.nseq, g0);
            u64 rend = vol.seq_off[r + 1];
#pragma unroll
            for (int i = 0; i < kPosPerThread; ++i) {
                const u64 p = g0 + i;
                if (p >= p1) break;
                while (p >= rend) { ++r; rend = vol.seq_off[r + 1]; }
                if (p + (u64)k <= rend) {
                    const u64 h = kmer_hash_at(vol.bases, (i64)p, k);
                    const u32 b = (u32)(h >> shift);
                    if (b < b_lo || b >= b_hi) continue;
                    if (MODE == 0 || pass == 0) atomicAdd(&hist[b], 1u);
                    else part[base[b] + atomicAdd(&hist[b], 1u)] = (h << kOffsetBits) | p;
                }
            }
        }
        __syncthreads();
        if (MODE == 0) {
            for (u32 i = threadIdx.x; i < nb; i += kPartThreads) if (hist[i]) atomicAdd(&bucket_cnt[i], hist[i]);
        } else if (pass == 0) {
            // reserve this block's slice of every bucket region, then rank locally
            for (u32 i = threadIdx.x; i < nb; i += kPartThreads) {
                const u32 c = hist[i];
                base[i] = c ? atomicAdd(reinterpret_cast<unsigned long long*>(&bucket_cursor[i]), (unsigned long long)c) : 0ULL;
                hist[i] = 0;
            }
            __syncthreads();
        }
    }
}

// exclusive scan of the bucket histogram (nb <= 4096) -> bucket_start[nb + 1]; cursors start there
__global__ void __launch_bounds__(1024)
k_bucket_scan(const u32* __restrict__ bucket_cnt, u32 nb, u64* __restrict__ bucket_start, u64* __restrict__ bucket_cursor)
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
// of slice (b, j) in part2 (absolute), sub_start[nb * 64] = number of records
__global__ void __launch_bounds__(256)
k_subpart(const u64* __restrict__ part, const u64* __restrict__ bucket_start, u32 nb, u64* __restrict__ part2, u64* __restrict__ sub_start)
{
    __shared__ u32 hist[4][kSubs], base[4][kSubs];
    const u32 b = blockIdx.x;
    const u64 lo = bucket_start[b], hi = bucket_start[b + 1];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    hist[w][lane] = 0;
    __syncthreads();
    // wave w owns the 64-record chunks w, w + 4, ...: it counts and later scatters the same records
    for (u64 e0 = lo + (u64)w * 64; e0 < hi; e0 += 256) {
        const u64 e = e0 + lane;
        if (e < hi) atomicAdd(&hist[w][(u32)(part[e] >> (kOffsetBits + kSliceBits)) & (kSubs - 1)], 1u);
    }
    __syncthreads();
    if (w == 0) {
        const u32 c0 = hist[0][lane], c1 = hist[1][lane], c2 = hist[2][lane], c3 = hist[3][lane];
        const u32 tot = c0 + c1 + c2 + c3;
        u32 incl = tot;
        for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        const u32 ex = incl - tot;
        base[0][lane] = ex; base[1][lane] = ex + c0; base[2][lane] = ex + c0 + c1; base[3][lane] = ex + c0 + c1 + c2;
        sub_start[(u64)b * kSubs + lane] = lo + ex;
        if (b == nb - 1 && lane == 0) sub_start[(u64)nb * kSubs] = hi;
    }
    __syncthreads();
    for (u64 e0 = lo + (u64)w * 64; e0 < hi; e0 += 256) {
        const u64 e = e0 + lane;
        if (e < hi) {
            const u64 rec = part[e];
            part2[lo + atomicAdd(&base[w][(u32)(rec >> (kOffsetBits + kSliceBits)) & (kSubs - 1)], 1u)] = rec;
        }
    }
}

template <int T = 256>
NECAT_D void slice_count(const u64* __restrict__ part2, u64 lo, u64 hi, u32* cnt)
{
    for (int i = threadIdx.x; i < kSlice; i += T) cnt[i] = 0;
    __syncthreads();
    for (u64 e = lo + threadIdx.x; e < hi; e += T) atomicAdd(&cnt[(u32)(part2[e] >> kOffsetBits) & (kSlice - 1)], 1u);
    __syncthreads();
}

// kept_tot[s] = number of offset-list entries slice s contributes (k-mers with 1..max_occ occurrences)
// (+ the same summed per bucket, so that the scan that follows runs over <= 4096 values, not 262 144)
__global__ void __launch_bounds__(256)
k_slice_count(const u64* __restrict__ part2, const u64* __restrict__ sub_start, u32 max_occ, u32* __restrict__ kept_tot, u32* __restrict__ bucket_kept, u32 s0)
{
    __shared__ u32 cnt[kSlice];
    __shared__ u32 red[4];
    const u64 s = (u64)blockIdx.x + s0;           // s0 = first slice of this rank's hash range
    slice_count(part2, sub_start[s], sub_start[s + 1], cnt);
    u32 sum = 0;
    for (int i = threadIdx.x; i < kSlice; i += 256) sum += filtered_count(cnt[i], max_occ);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { const u32 t = red[0] + red[1] + red[2] + red[3]; kept_tot[s] = t; if (t) atomicAdd(&bucket_kept[s >> kSubBits], t); }
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

// kmer_stats of the slice + its part of the offset list (tmp: same layout, order inside a k-mer not yet fixed)
template <int T>
__global__ void __launch_bounds__(T)
k_slice_emit(const u64* __restrict__ part2, const u64* __restrict__ sub_start, u32 max_occ, const u64* __restrict__ bucket_base,
             const u32* __restrict__ kept_tot, u64* __restrict__ kmer_stats, u32* __restrict__ tmp, u64* __restrict__ offset_list, u32 s0, u64 base_add)
{
    // s0 / base_add: first slice of this rank's hash range / offset-list entries of the ranks before it (the starts written
    // here are final: positions in the gathered list; tmp is addressed the same way by a pointer shifted back by base_add)
    __shared__ u32 cnt[kSlice];      // occurrences per table entry of the slice
    __shared__ u32 cur[kSlice];      // start of the entry's group inside the slice, then its fill cursor
    __shared__ u32 wtot[T / 64];
    __shared__ u64 s_base;
    const u64 s = (u64)blockIdx.x + s0;
    const u64 lo = sub_start[s], hi = sub_start[s + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {                 // the slice's base = its bucket's base + the kept totals of the bucket's earlier slices
        const u32 j = (u32)s & (kSubs - 1);
        u64 before = (u32)lane < j ? (u64)kept_tot[(s & ~(u64)(kSubs - 1)) + lane] : 0ULL;
        for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o);
        if (lane == 0) s_base = bucket_base[s >> kSubBits] + before + base_add;
    }
    __syncthreads();
    const u64 base = s_base;
    slice_count<T>(part2, lo, hi, cnt);
    // exclusive scan of the kept counts: thread t owns entries [E t, E t + E)
    constexpr int E = kSlice / T;
    u32 c[E], sum = 0;
#pragma unroll
    for (int i = 0; i < E; ++i) { c[i] = filtered_count(cnt[threadIdx.x * E + i], max_occ); sum += c[i]; }
    u32 incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    u32 run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wtot[w];
#pragma unroll
    for (int i = 0; i < E; ++i) { cur[threadIdx.x * E + i] = run; run += c[i]; }
    __syncthreads();
    // kmer_stats[h] = cnt<<34 | start, 0 for absent / over-represented k-mers (lookup_table.c:43-51, :94-113)
    u64* stats = kmer_stats + s * kSlice;
    for (int i = threadIdx.x; i < kSlice; i += T) {
        const u32 k = filtered_count(cnt[i], max_occ);
        stats[i] = k ? ((u64)k << kOffsetBits) | (base + cur[i]) : 0ULL;
    }
    __syncthreads();
    for (u64 e = lo + threadIdx.x; e < hi; e += T) {
        const u64 rec = part2[e];
        const u32 h = (u32)(rec >> kOffsetBits) & (kSlice - 1);
        if (filtered_count(cnt[h], max_occ)) tmp[base + atomicAdd(&cur[h], 1u)] = (u32)(rec & kOffsetMask);
    }
    __syncthreads();        // cur[h] is now the END of the group; the tmp writes of this workgroup are visible to it
    // radix_sort is stable (hash_list_bucket_sort.c:134): offsets ascend inside a k-mer
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
