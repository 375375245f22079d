// seed_kernels.h - __global__ shells around seed_core.h.
//   k_seed_hits   : one wave per query read; lanes stride over the sampled k-mers of both strands and
//                   sum their occurrence counts.  Random 8-byte gathers into kmer_stats (HBM/L2
//                   latency bound; SURVEY.md §8d "seeding" row).  The totals bound every per-read
//                   scratch size, so the second kernel never reallocates.
//   k_seed_reads  : one lane per query read, replaying the reference's order-dependent seeding state
//                   machine on a sparse per-lane block table (seed_core.h).  Reads are visited in
//                   descending hit-count order so the lanes of a wave carry similar work.
//   k_pack_cands  : compaction of the per-read outputs into one array of necat_candidate (global ids).
#pragma once
#include "seed_core.h"

namespace necat {

struct SeedMeta {
    u64 ht_off;      // entries
    u64 pool_off;    // SBlocks
    u64 chain_off;   // entries of (H+1)
    u64 out_off;     // DevCands
    u32 ht_mask, pool_cap, cs_cap, out_cap;
};

__global__ void __launch_bounds__(256)
k_seed_hits(DevVolume reads, const u64* __restrict__ kmer_stats, int k, int z, u32 read_lo, u32 read_hi, u32* __restrict__ hits)
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
        hf += (u32)(kmer_stats[rev2(xf) >> (64 - 2 * k)] >> kOffsetBits);
        hr += (u32)(kmer_stats[rev2(xr) >> (64 - 2 * k)] >> kOffsetBits);
    }
    for (int o = 32; o > 0; o >>= 1) { hf += __shfl_down(hf, o); hr += __shfl_down(hr, o); }
    if (lane == 0) { hits[2 * (u64)r] = hf; hits[2 * (u64)r + 1] = hr; }
}

struct SeedArenas {
    i32* ht_key; i32* ht_val; SBlock* pool;
    u64* cs; i32* f; i32* p; i32* t; i32* v; u64* u; DevCand* lcan;
    DevCand* out;
};

__global__ void __launch_bounds__(64)
k_seed_reads(DevVolume ref, DevVolume reads, const u64* __restrict__ kmer_stats, const u64* __restrict__ offset_list,
             SeedParams P, const u32* __restrict__ order, const SeedMeta* __restrict__ meta, u32 n,
             SeedArenas A, i32* __restrict__ n_cands, int* __restrict__ err_flag)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 r = order[i];
    const SeedMeta m = meta[i];
    SeedScratch S;
    S.ht_key = A.ht_key + m.ht_off; S.ht_val = A.ht_val + m.ht_off; S.ht_mask = m.ht_mask;
    S.pool = A.pool + m.pool_off; S.pool_cap = m.pool_cap;
    S.cs = A.cs + m.chain_off; S.f = A.f + m.chain_off; S.p = A.p + m.chain_off; S.t = A.t + m.chain_off;
    S.v = A.v + m.chain_off; S.u = A.u + m.chain_off; S.lcan = A.lcan + m.chain_off; S.cs_cap = m.cs_cap;
    S.out = A.out + m.out_off; S.out_cap = m.out_cap;
    const int nc = seed_one_read(ref, kmer_stats, offset_list, reads, (int)r, P, S);
    if (nc < 0) { atomicExch(err_flag, 1); n_cands[i] = 0; return; }
    n_cands[i] = nc;
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
