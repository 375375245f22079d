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

// MODE 0: count occurrences.  MODE 1: scatter offsets into bucket slots.
template <int MODE>
__global__ void __launch_bounds__(256)
k_kmer_pass(DevVolume vol, int k, u32* __restrict__ cnt32, const u64* __restrict__ kmer_stats, u64* __restrict__ tmp_list)
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
                    const u64 st = kmer_stats[h];
                    if (st >> kOffsetBits) {
                        const u32 old = atomicSub(&cnt32[h], 1u);
                        tmp_list[(st & kOffsetMask) + old - 1] = p;
                    }
                }
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
// present hashes get their start OR-ed in: lookup_table.c:94-113)
__global__ void __launch_bounds__(256)
k_write_stats(u32* __restrict__ cnt32, u64 n, u32 max_occ, const u64* __restrict__ partial, u64* __restrict__ kmer_stats)
{
    __shared__ u64 sh[256];
    const u64 base = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x * 8;
    u32 c[8]; u64 s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const u64 idx = base + i; c[i] = idx < n ? filtered_count(cnt32[idx], max_occ) : 0u; s += c[i]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 256 thread sums
    for (int off = 1; off < 256; off <<= 1) {
        u64 v = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    u64 run = partial[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u64 idx = base + i;
        if (idx < n) {
            kmer_stats[idx] = ((u64)c[i] << kOffsetBits) | (c[i] ? run : 0ULL);
            cnt32[idx] = c[i];       // becomes the bucket cursor of the scatter pass
            run += c[i];
        }
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
