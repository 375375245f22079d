// pcan_kernels.h - the candidate partitioner of the consensus stage (partition_candidates/pcan.c:39-103) for candidates that are
// still in this process: every candidate is offered as it is - template = its subject - and with the roles
// exchanged (change_pcan_roles, common/gapped_candidate.c:54-69); partition i holds the records whose template id lies in
// [i * batch_size, (i + 1) * batch_size).  Output = 28-byte PackedGappedCandidate records grouped by partition (the order inside a
// partition is as free as in the reference, whose worker threads append chunks).
#pragma once
#include "dev_common.h"
#include "../../include/necat_hip.h"

namespace necat {

struct PackedCan { u32 w[7]; };
static_assert(sizeof(PackedCan) == 28, "PackedGappedCandidate, common/gapped_candidate.h:64-66");

// pack_candidate (common/gapped_candidate.c:13-30)
NECAT_D PackedCan pcan_pack(const necat_candidate& c)
{
    PackedCan r;
    r.w[0] = (u32)(c.score < 1000000 ? c.score : 1000000) | (c.sdir == 1 ? 1u << 31 : 0u) | (c.qdir == 1 ? 1u << 30 : 0u) | (c.qoff == c.qbeg ? 1u << 29 : 0u);
    r.w[1] = (u32)c.sid; r.w[2] = (u32)c.sbeg; r.w[3] = (u32)c.send;
    r.w[4] = (u32)c.qid; r.w[5] = (u32)c.qbeg; r.w[6] = (u32)c.qend;
    return r;
}
// change_pcan_roles: the strand flags swap places, the anchor flag and the score stay, ids and ranges swap
NECAT_D PackedCan pcan_swap(const PackedCan& s)
{
    PackedCan d;
    d.w[0] = (s.w[0] & ((1u << 30) - 1)) | ((s.w[0] >> 31) << 30) | (((s.w[0] >> 30) & 1u) << 31);
    d.w[1] = s.w[4]; d.w[2] = s.w[5]; d.w[3] = s.w[6];
    d.w[4] = s.w[1]; d.w[5] = s.w[2]; d.w[6] = s.w[3];
    return d;
}
// partition of a template id, or -1 when the id is outside [0, num_parts * batch_size)  (pcan.c:47-75 tests the id against the
// range of every group of open files: an id outside all of them goes nowhere)
NECAT_D int pcan_part(i32 id, int batch_size, int nparts)
{
    if (id < 0) return -1;
    const int p = id / batch_size;
    return p < nparts ? p : -1;
}

constexpr int kPcanLds = 2048;      // partitions counted in LDS per block (more: global atomics per record)

// MODE 0: records per partition.  MODE 1: scatter; cursor[p] runs from the partition's start.
template <int MODE>
__global__ void __launch_bounds__(256)
k_pcan(const necat_candidate* __restrict__ cands, u64 n, int batch_size, int nparts, unsigned long long* __restrict__ cursor, PackedCan* __restrict__ out)
{
    __shared__ u32 cnt[kPcanLds];
    __shared__ unsigned long long base[MODE == 1 ? kPcanLds : 1];
    const bool lds = nparts <= kPcanLds;
    if (lds) { for (int i = threadIdx.x; i < nparts; i += 256) cnt[i] = 0; __syncthreads(); }
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    int ps = -1, pq = -1; u32 rs = 0, rq = 0;
    PackedCan r;
    if (i < n) {
        const necat_candidate c = cands[i];
        r = pcan_pack(c);
        ps = pcan_part(c.sid, batch_size, nparts); pq = pcan_part(c.qid, batch_size, nparts);
    }
    if (lds) {
        if (ps >= 0) rs = atomicAdd(&cnt[ps], 1u);
        if (pq >= 0) rq = atomicAdd(&cnt[pq], 1u);
        __syncthreads();
        for (int p = threadIdx.x; p < nparts; p += 256) {
            const u32 c = cnt[p];
            if (!c) continue;
            const unsigned long long b = atomicAdd(&cursor[p], (unsigned long long)c);
            if (MODE == 1) base[p] = b;
        }
        if (MODE == 1) {
            __syncthreads();
            if (ps >= 0) out[base[ps] + rs] = r;
            if (pq >= 0) out[base[pq] + rq] = pcan_swap(r);
        }
    } else {
        if (ps >= 0) { const unsigned long long at = atomicAdd(&cursor[ps], 1ULL); if (MODE == 1) out[at] = r; }
        if (pq >= 0) { const unsigned long long at = atomicAdd(&cursor[pq], 1ULL); if (MODE == 1) out[at] = pcan_swap(r); }
    }
}

}  // namespace necat
