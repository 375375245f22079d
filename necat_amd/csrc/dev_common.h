// dev_common.h - shared host/device definitions for the gfx950 overlap kernels.
//
// The per-lane "core" routines of every kernel are written as NECAT_HD inline functions so that
// tests/host_core can compile exactly the same source with g++ and run it lane-by-lane against
// the oracle on a machine without a GPU.  Only the __global__ shells (grid mapping, wave
// intrinsics, memory layout strides) are HIP-only.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NECAT_HD __host__ __device__ __forceinline__
#define NECAT_D __device__ __forceinline__
#else
#define NECAT_HD inline
#define NECAT_D inline
#endif

namespace necat {

typedef uint64_t u64;
typedef uint32_t u32;
typedef int64_t i64;
typedef int32_t i32;
typedef int16_t i16;
typedef uint16_t u16;
typedef uint8_t u8;

constexpr int kOffsetBits = 34;                         // lookup_table.h:12
constexpr u64 kOffsetMask = (1ULL << kOffsetBits) - 1;  // lookup_table.h:15
constexpr int kBlkSeeds = 40;                           // word_finder_aux.h:9
constexpr int kOcaBlockSize = 512;                      // edlib_ex_aux.h:23
constexpr int kOcaMatCnt = 8;                           // oc_aligner.c:9
constexpr int kMaxFragLen = 794;                        // oc_aligner.c:127-131: < 612 * 1.3
constexpr int kMaxWords = 13;                           // ceil(794 / 64)
constexpr int kMaxTWords = 25;                          // ceil(794 / 32)
constexpr int kWave = 64;

NECAT_HD int popc64(u64 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
NECAT_HD int ctz64(u64 x)  // x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}

#if defined(__HIPCC__)
// ---- wave-wide sums without the LDS crossbar.  __shfl_up / __shfl_down are ds_bpermute (an LDS instruction + its address arithmetic + a select
// per step); DPP moves ride on the add itself.  Inclusive prefix sum over the 64 lanes: Kogge-Stone inside the rows of 16 (row_shr 1, 2, 4, 8:
// lanes without a source add 0), then lane 15 of every row into the next row (rows 1 and 3), then lane 31 into rows 2 and 3.
NECAT_D u32 wave_scan_add(u32 v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (u32)x;
}
// the wave's total from an inclusive scan (lane 63's value, on every lane: a scalar)
NECAT_D u32 wave_last(u32 incl) { return (u32)__builtin_amdgcn_readlane((int)incl, 63); }
// OR over each group of 8 consecutive lanes, on all 8: quad_perm [1,0,3,2], [2,3,0,1], then the mirror image inside the half row (the other quad)
NECAT_D u32 or_lanes8(u32 v)
{
    int x = (int)v;
    x |= __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);
    x |= __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);
    return (u32)x;
}
#endif
NECAT_HD u64 brev64(u64 x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    return __builtin_bswap64(x);
#endif
}

// ---------------------------------------------------------------------------------------------
// Device volume: bases re-packed on upload to little-endian 2-bit words: base i lives in bits
// [2*(i&31), 2*(i&31)+1] of 64-bit word i>>5 (the reference's pac keeps the first base of a byte in
// its TOP bits, ontcns_aux.h:118-119; volume.hip converts).  `bases` has 2 guard words before index
// 0 and 4 after the end so unaligned 64-base windows never fault.
// ---------------------------------------------------------------------------------------------
struct DevVolume {
    const u64* bases;     // 2-bit LE packed (guarded)
    const u64* seq_off;   // [nseq + 1], seq_off[nseq] = nbases
    u64 nbases;
    u64 nseq;
};

NECAT_HD int base_at(const u64* bases, i64 g)
{
    return (int)((bases[g >> 5] >> ((g & 31) * 2)) & 3);
}

// 32 bases g .. g+31 (ascending) as one 2-bit packed word; g may be unaligned and may be negative
// down to -64 (guard words).
NECAT_HD u64 load32(const u64* bases, i64 g)
{
    i64 w = g >> 5;          // arithmetic shift: floor
    int sh = (int)(g & 31) * 2;
    u64 lo = bases[w];
    if (sh == 0) return lo;
    u64 hi = bases[w + 1];
    return (lo >> sh) | (hi << (64 - sh));
}

// reverse the order of the 32 two-bit groups of x
NECAT_HD u64 rev2(u64 x)
{
    x = brev64(x);
    return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
}

// fragment element i (0..31) of a window: ascending (dir=+1) element i = base(g0 + i),
// descending (dir=-1) element i = base(g0 - i); comp => 3 - code.
NECAT_HD u64 load32_dir(const u64* bases, i64 g0, int dir, int comp)
{
    u64 x = dir > 0 ? load32(bases, g0) : rev2(load32(bases, g0 - 31));
    return comp ? ~x : x;
}

// even bits of x compacted into the low 32 bits
NECAT_HD u64 even_bits(u64 x)
{
    x &= 0x5555555555555555ULL;
    x = (x | (x >> 1)) & 0x3333333333333333ULL;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFULL;
    return x;
}

// 64 fragment elements starting at element e0 -> two bit-planes (bit r = element e0 + r)
NECAT_HD void load64_planes(const u64* bases, i64 g0, int dir, int comp, int e0, u64* lo, u64* hi)
{
    u64 a = load32_dir(bases, g0 + (i64)dir * e0, dir, comp);
    u64 b = load32_dir(bases, g0 + (i64)dir * (e0 + 32), dir, comp);
    *lo = even_bits(a) | (even_bits(b) << 32);
    *hi = even_bits(a >> 1) | (even_bits(b >> 1) << 32);
}

// The k-mer table as the kernels read it.  Dense: kmer_stats[h] = cnt << 34 | start, the reference layout (4^k entries,
// 8.6 GB at k = 15, 82 % of them zero at E. coli size).  Sparse (what the LDS-slice build writes): one IdxWord per 64 table
// entries - which of them are non-zero and where the first of those sits in `compact`, the non-zero entries in hash order.
// A lookup is a 16-byte load + (for a k-mer that exists) an 8-byte load; 0.27 + 1.5 GB instead of 8.6 GB to write, to
// exchange between ranks and to hold.
struct __attribute__((aligned(16))) IdxWord { u64 bits, base; };
struct IndexView {
    const u64* dense;
    const IdxWord* words;
    const u64* compact;
    NECAT_HD u64 lookup(u64 h) const
    {
        if (dense) return dense[h];
        const IdxWord w = words[h >> 6];
        const u64 bit = 1ULL << (h & 63);
        if (!(w.bits & bit)) return 0ULL;
        return compact[w.base + (u64)popc64(w.bits & (bit - 1))];
    }
};

// lower-bound style search: id of the sequence containing global offset g (packed_db.c:173-189
// returns the same id for every in-range offset).
NECAT_HD u64 seq_of_offset(const u64* seq_off, u64 nseq, u64 g)
{
    u64 lo = 0, hi = nseq;       // invariant: seq_off[lo] <= g < seq_off[hi]
    while (hi - lo > 1) {
        u64 mid = (lo + hi) >> 1;
        if (seq_off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

}  // namespace necat
