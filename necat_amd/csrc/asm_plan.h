// asm_plan.h - oc2asmpm's candidate stage on the device (the host statement of the same steps is asm_core.h, which the CPU
// tests pin to the reference's own oc2asmpm):
//
//   vote     pairwise_mapping (asm_pm/asm_pm_common.c:509-702) per (read, strand): every BC-th k-mer looked up in the volume's table, its hits
//            binned into 1000-bp blocks of the reference volume (at most 60 per block, first hit of a k-mer per block), the touched blocks
//            visited in first-touch order: find_location (:479-507, the all-pairs distance vote), one candidate per passing block, topped up
//            with the agreeing hits of the blocks to either side, which a candidate uses up.
//            k_seed_hits (seed_kernels.h, with z = BC) sizes the scratch and keeps the table words;
//            k_asm_vote_collect  one wave per (read, strand): the ordered walk flattened into chunks of 64 consecutive hits in which lane
//                                order is sequence order (the machinery of k_seed_collect_wave with this stage's rules: every first hit
//                                updates the block's stale pair score, a full block still counts);
//            k_asm_vote_eval     one wave per (read, strand): the blocks in first-touch order, the O(n^2) vote one row per lane, the
//                                top-up as one flat sequence of the neighbour blocks' hits;
//            k_asm_select        one wave per read: AsmGappedCandidate_ScoreGT (:145-153) by ranks, the first -n candidates, one entry per
//                                (subject, strand) in that order (asm_core.h BatchMapper::plan explains why that is the walk's outcome).
//   range    compute_align_range_1 (asm_pm/find_mem.c:222-265) per planned (read, subject strand): 10-mers of the subject at every 6th
//            position against all 10-mers of the read (find_kmer_match :92-133 with its occurrence limits; both lists' first entries carry hash
//            0, :22-74), each match extended to a maximal exact match, the matches an earlier one covers dropped (:173-220), the rest (>= 15
//            bp) chained (scoring_mems, asm_pm/km_chain.c:141-205) and the best chain's middle match made the anchor (mem_find_best_can :322-445).
//            k_asm_subj_occ      per (subject, strand): how often each sampled 10-mer occurs among the sampled ones (a counting table in a
//                                per-wave scratch), one byte per sampled position;
//            k_asm_read_index    per read: open-addressing table 10-mer -> chain of its positions (built once per read, as the reference
//                                sorts the read's list once per read, asm_pm_common.c:375-383);
//            k_asm_seeds<false>  per pair: number of (subject position, read position) matches that pass the limits (sizes the scratch);
//            k_asm_seeds<true>   per pair: the matches in subject-position order, each extended word-wise on the 2-bit volumes; a match is
//                                dropped iff the nearest earlier match on its diagonal reaches it (that is what the reference's in-order
//                                sweep over the sorted matches does: matches of one diagonal are met in ascending order, and an earlier one
//                                covers a later one exactly when its right end lies beyond the later one's start); survivors of >= 15 bp
//                                are rank-sorted by (read offset, subject offset);
//            k_asm_chain         per pair: the chain DP with the predecessor scan of a match taken 64 candidates at a time (chain_fill_wave's
//                                prefix formulation of max_skip), the best chain end by a wave reduction, the anchor.
#pragma once
#include "seed_kernels.h"

namespace necat {

constexpr int kAsmZV = 1000, kAsmSM = 60;                 // ZV, SM (asm_pm_common.c:22,24)
constexpr int kAsmIdxCut = 5, kAsmBlkCut = 4;             // KMER_CNT_CUTOFF, BLOCK_SCORE_CUTOFF (:32-33)
constexpr int kAsmRangeK = 10, kAsmRangeW = 6, kAsmMinMem = 15, kAsmMaxOcc = 20;      // asm_pm_common.c:346,375-383; find_mem.c:237

struct VBlock {              // Back_List (:449-452) + its index_list / index_score entry
    i32 score, stale, block_id, slot;
    i32 seedno[kAsmSM];
    i16 loc[kAsmSM];
    i32 _pad[2];
};
static_assert(sizeof(VBlock) == 384, "VBlock layout");
struct VoteCand { i32 score, chain, target_id, query_start, target_start, target_size; };      // AsmGappedCandidate (:119-125) without readno
struct VoteMeta { u64 ht_off[2], pool_off[2]; u32 ht_mask[2], pool_cap[2]; };                  // per processed read; candidates of a strand go to out[pool_off ..]
struct VoteArenas { u64* ht; VBlock* pool; VoteCand* out; };
struct VoteParams { int k, bc, read_start_id, ref_start_id, num_extended; };
struct AsmPlanDev { i32 sid, sdir, qoff, soff, score, ssize; };          // qoff < 0: no chain

NECAT_D VBlock* vb_find(const u64* ht, u32 mask, VBlock* pool, i32 block_id)
{
    if (block_id < 0) return nullptr;
    u32 h = ht_hash(block_id, mask);
    for (;;) {
        const u64 e = ht[h];
        if ((i32)(u32)e == block_id && e != kHtEmpty) return pool + (u32)(e >> 32);
        if (e == kHtEmpty) return nullptr;
        h = (h + 1) & mask;
    }
}

// find_location's test (:483): quotient and "- 1" in float, fabs and the comparison in double.  Callers pass dloc > 0, dseed > 0.
// Without the division, exactly: with D = dseed * len (an integer below 2^24, so the float product and both operands are exact) the quotient is the
// correctly rounded float of dloc / D; r - 1.0f is exact near 1; |r - 1| < 0.1 (the double) holds for the floats 0x3F666667 .. 0x3F8CCCCC, i.e. for
// dloc / D above the midpoint 0.9 + 5.96e-9 (a tie rounds to the even 0x3F666666, which fails) and up to the midpoint 1.1 - 3.58e-8 (a tie rounds to the
// even 0x3F8CCCCC, which passes).  10 dloc and 9 D, 11 D are integers and the two offsets times 10 D stay far below 1, so that is 9 D < 10 dloc < 11 D.
NECAT_D bool asm_ratio_ok(int dloc, int dseed, float len)
{
    const int li = (int)len;
    if ((float)li == len && li > 0 && dloc < (1 << 24) && (i64)dseed * li < (1 << 24)) {
        const i64 a = 10 * (i64)dloc, d = (i64)dseed * li;
        return a > 9 * d && a < 11 * d;
    }
    const float r = (float)dloc / ((float)dseed * len);
    double d = (double)(r - 1.0f);
    if (d < 0) d = -d;
    return d < 0.10;
}
// the top-up's test (:685, :693): fabs(x / (y * 1.0) - 1.0) < 0.10, all in double, x and y any ints.  Exactly, without the division: the quotient is the
// correctly rounded double of x / y; the test holds for the doubles 0x3FECCCCCCCCCCCCD .. 0x3FF1999999999999, i.e. for x / y >= 0.9 - 3.3e-17 (the double
// nearest to 0.9 passes) and < 1.1 - 2.2e-17 (the double nearest to 1.1 fails); in integers 9 |y| <= 10 |x| < 11 |y| with x, y of one sign, y != 0.
NECAT_D bool asm_ratio_ok_f64(int x, int y)
{
    if (y == 0 || x == 0 || ((x < 0) != (y < 0))) return false;
    const i64 ax = x < 0 ? -(i64)x : (i64)x, ay = y < 0 ? -(i64)y : (i64)y;
    return 9 * ay <= 10 * ax && 10 * ax < 11 * ay;
}

// ---- collection: k_seed_collect_wave with this stage's rules
__global__ void __launch_bounds__(64)
k_asm_vote_collect(DevVolume ref, DevVolume reads, IndexView index, const u64* __restrict__ offset_list, VoteParams P, const u32* __restrict__ order,
                   const VoteMeta* __restrict__ meta, u32 n, VoteArenas A, i32* __restrict__ nblk_out, int* __restrict__ err_flag, const u64* __restrict__ kst)
{
    __shared__ u32 s_pre[65];
    __shared__ u64 s_list[64];
    const u32 t = blockIdx.x;
    if (t >= 2 * n) return;
    const u32 i = t >> 1;
    const int strand = (int)(t & 1);
    const int lane = threadIdx.x;
    const u64 below = (1ULL << lane) - 1ULL;
    const VoteMeta m = meta[i];
    u64* const ht = A.ht + m.ht_off[strand];
    const u32 ht_mask = m.ht_mask[strand];
    VBlock* const pool = A.pool + m.pool_off[strand];
    const u32 pool_cap = m.pool_cap[strand];
    const int read_id = (int)order[i];
    const u64 q_goff = reads.seq_off[read_id];
    const int L = (int)(reads.seq_off[read_id + 1] - q_goff);
    u64 soff_max = ~0ULL;                       // :539-543
    {
        const int gid = read_id + P.read_start_id;
        if (gid >= P.ref_start_id && gid < P.ref_start_id + (int)ref.nseq) soff_max = ref.seq_off[gid - P.ref_start_id];      // (the host Voter's ref_off[gid - ref_start]: the read's own copy in the reference volume, whatever the two start ids)
    }
    const int k = P.k, z = P.bc;
    const int nk = L >= k ? (L - k) / z + 1 : 0;
    int nblk = 0;
    bool failed = false;
    for (int kbase = 0; kbase < nk && !failed; kbase += 64) {
        const int kj = kbase + lane;
        u32 cnt = 0; u64 lst = 0;
        if (kj < nk) {
            u64 st;
            if (kst) st = kst[seed_kst_base(q_goff, (u32)read_id, z) + 2 * (u64)kj + (u64)strand];
            else {
                const int pos = kj * z;
                const u64 x = strand == 0 ? load32_dir(reads.bases, (i64)q_goff + pos, +1, 0) : load32_dir(reads.bases, (i64)q_goff + L - 1 - pos, -1, 1);
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
        __syncthreads();
        s_pre[lane] = inc - cnt; s_list[lane] = lst;
        if (lane == 63) s_pre[64] = inc;
        __syncthreads();
        const u32 T = s_pre[64];
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
                const u64 q = off / (u64)kAsmZV;
                blk = (i32)q; boff = (int)(off - q * (u64)kAsmZV);
                cand = true;
                if (kk > 0) cand = !(offp >= q * (u64)kAsmZV);        // the k-mer's previous hit lies in the same block: B.seednum == k + 1 (:572)
            }
            const i32 kmer_id = kbase + j + 1;
            const u64 cmask = __ballot(cand);
            int rank = 0, first = lane, later = 0, cprev = 0;
            for (u64 todo = cmask; todo;) {
                const int l = ctz64(todo);
                const i32 bl = __builtin_amdgcn_readlane(blk, l);
                const bool mine = cand && blk == bl;
                const u64 same = __ballot(mine);
                if (mine) { rank = popc64(same & below); later = popc64(same >> lane) - 1; first = l; }
                if (cand && blk == bl + 1) cprev = popc64(same & below);
                todo &= ~same;
            }
            const bool is_first = cand && rank == 0;
            i32 idx = -1, ip = -1; u32 h = 0;
            {
                u32 hp = 0;
                const bool want_own = is_first, want_prev = cand && blk > 0;
                u64 e_own = kHtEmpty, e_prev = kHtEmpty;
                if (want_own) { h = ht_hash(blk, ht_mask); e_own = __hip_atomic_load(&ht[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                if (want_prev) { hp = ht_hash(blk - 1, ht_mask); e_prev = __hip_atomic_load(&ht[hp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                if (want_own) {
                    while (e_own != kHtEmpty && (i32)(u32)e_own != blk) { h = (h + 1) & ht_mask; e_own = __hip_atomic_load(&ht[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                    idx = e_own == kHtEmpty ? -1 : (i32)(u32)(e_own >> 32);
                }
                if (want_prev) {
                    while (e_prev != kHtEmpty && (i32)(u32)e_prev != blk - 1) { hp = (hp + 1) & ht_mask; e_prev = __hip_atomic_load(&ht[hp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                    ip = e_prev == kHtEmpty ? -1 : (i32)(u32)(e_prev >> 32);
                }
            }
            const bool create = is_first && idx < 0;
            const u64 crm = __ballot(create);
            const int ncreate = popc64(crm);
            if ((u32)(nblk + ncreate) > pool_cap) { failed = true; break; }
            if (create) {
                idx = nblk + popc64(crm & below);
                for (;;) {
                    const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&ht[h]), (unsigned long long)kHtEmpty, (unsigned long long)ht_entry(blk, idx));
                    if (old == (unsigned long long)kHtEmpty) break;
                    h = (h + 1) & ht_mask;
                }
                VBlock* vb = pool + idx;
                vb->score = 0; vb->stale = 0; vb->block_id = blk; vb->slot = (i32)h;
            }
            nblk += ncreate;
            idx = __shfl(idx, first);
            __syncthreads();
            int s0 = 0, s0p = 0;
            if (cand) s0 = pool[idx].score;
            if (cand && ip >= 0) s0p = pool[ip].score;
            // every first hit counts (:573-580): it takes slot score while that is below 60, the score stops at 60, and the pair score kept with the
            // block's first-touch entry is rewritten each time - with the neighbour's score as of then
            const int at = s0 + rank;
            const bool is_last = cand && later == 0;
            int after = at + 1; if (after > kAsmSM) after = kAsmSM;
            int sprev = s0p + cprev; if (sprev > kAsmSM) sprev = kAsmSM;
            __syncthreads();
            if (cand) {
                VBlock* vb = pool + idx;
                if (at < kAsmSM) { vb->loc[at] = (i16)boff; vb->seedno[at] = kmer_id; }
                if (is_last) { vb->score = after; vb->stale = after + sprev; }
            }
            __syncthreads();
        }
    }
    if (lane == 0) {
        if (failed) atomicExch(err_flag, 1);
        nblk_out[t] = failed ? 0 : nblk;
    }
}

__global__ void __launch_bounds__(64)
k_asm_vote_clear(const VoteMeta* __restrict__ meta, u32 n, VoteArenas A, const i32* __restrict__ nblk_in)
{
    const u32 t = blockIdx.x;
    if (t >= 2 * n) return;
    const VoteMeta m = meta[t >> 1];
    const int st = (int)(t & 1);
    u64* ht = A.ht + m.ht_off[st];
    const VBlock* pool = A.pool + m.pool_off[st];
    const int nb = nblk_in[t];
    for (int b = (int)threadIdx.x; b < nb; b += 64) ht[pool[b].slot] = kHtEmpty;
}

// the votes row i collects from / gives to the rows j > i (:483)
template <class AddJ>
NECAT_D int asm_vote_row(const int* t_loc, const int* t_seedn, int i, int k, float len, int read_len, AddJ& add_j)
{
    int own = 0, last = t_seedn[i];
    for (int j = i + 1; j < k; ++j)
        if (last != t_seedn[j] && t_seedn[j] - t_seedn[i] > 0 && t_loc[j] - t_loc[i] > 0 && t_loc[j] - t_loc[i] < read_len &&
            asm_ratio_ok(t_loc[j] - t_loc[i], t_seedn[j] - t_seedn[i], len)) { ++own; add_j(j); last = t_seedn[j]; }
    return own;
}

// the second half of find_location (:485-505) on the lanes of a wave (k <= 120: indices lane and lane + 64); as in the seeding stage's
// scoring_pick_wave the anchor of the "rep != maxval" walk is its first element with a non-zero offset, or its last element
NECAT_D int asm_pick_wave(const int* s_loc, const int* s_seedn, const int* s_score, int k, float len, int read_len, int lane, int* msid)
{
    const int a0 = lane, a1 = lane + 64;
    const int v0 = a0 < k ? s_score[a0] : -1, v1 = a1 < k ? s_score[a1] : -1;
    const int maxval = wave_read(wave_prefix_max(v0 > v1 ? v0 : v1), 63);
    if (maxval < 5) return 0;
    const u64 e0 = __ballot(v0 == maxval), e1 = __ballot(v1 == maxval);
    const int maxi = e0 ? ctz64(e0) : 64 + ctz64(e1);
    const int rep = popc64(e0) + popc64(e1) - 1;
    if (rep == maxval) { *msid = maxi; return 1; }
    const int lm = s_loc[maxi], sm = s_seedn[maxi];
    auto agrees = [&](int j) {
        if (j >= k || j == maxi) return j == maxi;
        const int lj = s_loc[j], sj = s_seedn[j];
        if (j < maxi) return sm - sj > 0 && lm - lj > 0 && lm - lj < read_len && asm_ratio_ok(lm - lj, sm - sj, len);
        return sj - sm > 0 && lj - lm > 0 && lj - lm <= read_len && asm_ratio_ok(lj - lm, sj - sm, len);
    };
    const bool g0 = agrees(a0), g1 = agrees(a1);
    const u64 w0 = __ballot(g0), w1 = __ballot(g1);
    const u64 n0 = __ballot(g0 && s_loc[a0 < k ? a0 : 0] != 0), n1 = __ballot(g1 && s_loc[a1 < k ? a1 : 0] != 0);
    int pick;
    if (n0) pick = ctz64(n0);
    else if (n1) pick = 64 + ctz64(n1);
    else pick = w1 ? 64 + 63 - __clzll((long long)w1) : 63 - __clzll((long long)w0);
    *msid = pick;
    return 1;
}

__global__ void __launch_bounds__(64)
k_asm_vote_eval(DevVolume ref, DevVolume reads, VoteParams P, const u32* __restrict__ order, const VoteMeta* __restrict__ meta, u32 n, VoteArenas A,
                const i32* __restrict__ nblk_in, i32* __restrict__ n_strand)
{
    __shared__ int s_loc[128], s_seedn[128], s_score[128];
    __shared__ int s_pre[65], s_rel[64];
    __shared__ VBlock* s_vb[64];
    const u32 t = blockIdx.x;
    if (t >= 2 * n) return;
    const u32 i = t >> 1;
    const int strand = (int)(t & 1);
    const int lane = threadIdx.x;
    const u64 below = (1ULL << lane) - 1ULL;
    const VoteMeta m = meta[i];
    const u64* const ht = A.ht + m.ht_off[strand];
    const u32 ht_mask = m.ht_mask[strand];
    VBlock* const pool = A.pool + m.pool_off[strand];
    VoteCand* const out = A.out + m.pool_off[strand];
    const int r = (int)order[i];
    const int L = (int)(reads.seq_off[r + 1] - reads.seq_off[r]);
    const int gid = r + P.read_start_id;
    const int nblk = nblk_in[t];
    const float len = (float)P.bc;
    int n_out = 0;
    u64 pending = 0; int pbase = -64;
    for (;;) {
        __syncthreads();
        if (!pending) {
            // a block's score only ever drops and its stale score is fixed: 64 blocks are tested at once, the survivors again at their turn
            pbase += 64;
            if (pbase >= nblk) break;
            bool pass = false;
            if (pbase + lane < nblk) { const VBlock* q = pool + pbase + lane; pass = q->stale > kAsmIdxCut && q->score != 0; }
            pending = __ballot(pass);
            continue;
        }
        const int bi = pbase + ctz64(pending);
        pending &= pending - 1;
        VBlock* const B = pool + bi;
        if (!(B->stale > kAsmIdxCut && B->score != 0)) continue;
        const int b = B->block_id;
        // the hits of the left neighbour, if it has any, then the block's own shifted by one block (:588-601)
        VBlock* const Pv = b > 0 ? vb_find(ht, ht_mask, pool, b - 1) : nullptr;
        const int np = Pv ? Pv->score : 0, nc = B->score;
        const int start_loc = np > 0 ? (b - 1) * kAsmZV : b * kAsmZV;
        if (lane < np) { s_loc[lane] = Pv->loc[lane]; s_seedn[lane] = Pv->seedno[lane]; }
        if (lane < nc) { s_loc[np + lane] = B->loc[lane] + (np ? kAsmZV : 0); s_seedn[np + lane] = B->seedno[lane]; }
        const int ns = np + nc;
        for (int x = lane; x < 128; x += 64) s_score[x] = 0;
        __syncthreads();
        // rows 0 .. ns - 2, one or two per lane; the rows beyond 63 in reverse (row i scans ns - 1 - i hits: lane l gets rows l and ns - 2 - l, ns hits in all)
        for (int q = 0; q < 2; ++q) {
            const int ii = q == 0 ? lane : ns - 2 - lane;
            if (ii < ns - 1 && (q == 0 || ii >= 64)) {
                LdsAdder add; add.s = s_score;
                const int own = asm_vote_row(s_loc, s_seedn, ii, ns, len, L, add);
                if (own) atomicAdd(&s_score[ii], own);
            }
        }
        __syncthreads();
        int msid = -1;
        if (!asm_pick_wave(s_loc, s_seedn, s_score, ns, len, L, lane, &msid)) continue;
        const int score0 = s_score[msid];
        if (score0 < kAsmBlkCut) continue;
        const int loc_seed = s_seedn[msid], loc_list = s_loc[msid] + start_loc;
        const int readno = (int)seq_of_offset_wave(ref.seq_off, ref.nseq, (u64)loc_list, lane);
        const int readstart = (int)ref.seq_off[readno], length = (int)(ref.seq_off[readno + 1] - ref.seq_off[readno]), readend = readstart + length;
        if (readno + P.ref_start_id == gid) {
            // the read itself (:604-612; unreachable while hits at or beyond the read's own offset are skipped, :539-543): its stretch is wiped
            if (lane == 0) {
                int u = readstart / kAsmZV, s = readstart % kAsmZV, kk = 0;
                VBlock* T = vb_find(ht, ht_mask, pool, u);
                if (T) { for (int j = 0; j < T->score && j < kAsmSM; ++j) if (T->loc[j] < s) T->loc[kk++] = T->loc[j]; T->score = kk; }
                for (++u; u < readend / kAsmZV; ++u) { T = vb_find(ht, ht_mask, pool, u); if (T) T->score = 0; }
                kk = 0; s = readend % kAsmZV;
                T = vb_find(ht, ht_mask, pool, u);
                if (T) { for (int j = 0; j < T->score && j < kAsmSM; ++j) if (T->loc[j] > s) T->loc[kk++] = T->loc[j]; T->score = kk; }
            }
            continue;
        }
        const int qstart = (loc_seed - 1) * P.bc;
        const int left1 = loc_list - readstart + P.k - 1, right1 = readend - loc_list;
        const int left2 = qstart + P.k - 1, right2 = L - qstart;
        const int num1 = left1 >= left2 ? left2 : left1, num2 = right1 >= right2 ? right2 : right1;
        if (num1 + num2 < 400) continue;
        // ---- the agreeing hits of the blocks to either side (:630-647): slots [0, nleft) = blocks b - 2, b - 3, ..; [nleft, nslot) = b + 1, b + 2, ..
        int nleft = num1 / kAsmZV + 1; if (nleft > b - 1) nleft = b - 1; if (nleft < 0) nleft = 0;
        const int nslot = nleft + num2 / kAsmZV;
        int seedcount = 0;
        for (int s0 = 0; s0 < nslot; s0 += 64) {
            const int sl = s0 + lane;
            const int ub = sl < nleft ? b - 2 - sl : b + 1 + (sl - nleft);
            VBlock* T = sl < nslot ? vb_find(ht, ht_mask, pool, ub) : nullptr;
            const int nsc = T ? T->score : 0;
            const int inc = wave_prefix_sum(nsc);
            __syncthreads();
            s_pre[lane] = inc - nsc; s_rel[lane] = 0; s_vb[lane] = T;
            if (lane == 63) s_pre[64] = inc;
            __syncthreads();
            const int Tn = s_pre[64];
            for (int c0 = 0; c0 < Tn; c0 += 64) {
                const int q = c0 + lane;
                int lo = 0;
                for (int st = 32; st > 0; st >>= 1) if (s_pre[lo + st] <= q) lo += st;
                bool acc = false;
                if (q < Tn) {
                    const VBlock* X = s_vb[lo];
                    const int e = q - s_pre[lo], sl2 = s0 + lo;
                    const int u2 = sl2 < nleft ? b - 2 - sl2 : b + 1 + (sl2 - nleft);
                    const int at = u2 * kAsmZV;
                    const int hl = X->loc[e], hs = X->seedno[e];
                    if (sl2 < nleft) acc = asm_ratio_ok_f64(loc_list - at - hl, (loc_seed - hs) * P.bc);
                    else acc = asm_ratio_ok_f64(at + hl - loc_list, (hs - loc_seed) * P.bc);
                }
                const u64 mask = __ballot(acc);
                if (acc) atomicAdd(&s_rel[lo], 1);
                seedcount += popc64(mask);
            }
            __syncthreads();
            if (nsc && (double)s_rel[lane] * 1.0 / (double)nsc > 0.4) T->score = 0;
        }
        (void)below;
        if (lane == 0) {
            VoteCand c;
            c.score = score0 + seedcount; c.chain = strand; c.target_id = readno; c.query_start = qstart;
            c.target_start = loc_list - readstart; c.target_size = length;
            out[n_out] = c;
        }
        ++n_out;
    }
    if (lane == 0) n_strand[t] = n_out;
}

// AsmGappedCandidate_ScoreGT (:145-153)
NECAT_D bool vote_before_dev(const VoteCand& a, const VoteCand& b)
{
    if (a.score != b.score) return a.score > b.score;
    if (a.chain != b.chain) return a.chain < b.chain;
    if (a.target_id != b.target_id) return a.target_id < b.target_id;
    if (a.query_start != b.query_start) return a.query_start < b.query_start;
    return a.target_start < b.target_start;
}

// one wave per read: its candidates (forward strand's, then reverse strand's) ranked by the order above, the first num_extended of them, one plan
// entry per (subject, strand) in that order.  sel: num_extended VoteCands of scratch per read; plan: num_extended entries per read.
// The order as two ascending 64-bit keys (score descending = its complement ascending; the fields are non-negative ints): up to kSelLds
// candidates are ranked on keys kept in LDS, longer lists on the records in global memory.
constexpr int kSelLds = 4096;          // 64 KB of LDS: a read in a repeat has thousands of candidates, and ranking them in global memory made the launch's tail
NECAT_D void vote_keys(const VoteCand& c, u64* k1, u64* k2)
{
    *k1 = ((u64)(0x7fffffffu - (u32)c.score) << 32) | ((u64)(u32)c.chain << 31) | (u64)(u32)c.target_id;
    *k2 = ((u64)(u32)c.query_start << 32) | (u64)(u32)c.target_start;
}
__global__ void __launch_bounds__(64)
k_asm_select(VoteParams P, const VoteMeta* __restrict__ meta, u32 n, VoteArenas A, const i32* __restrict__ n_strand, VoteCand* __restrict__ sel,
             AsmPlanDev* __restrict__ plan, i32* __restrict__ nplan)
{
    __shared__ ulonglong2 l_key[kSelLds];
    const u32 i = blockIdx.x;
    if (i >= n) return;
    const int lane = threadIdx.x;
    const u64 below = (1ULL << lane) - 1ULL;
    const VoteMeta m = meta[i];
    const int n0 = n_strand[2 * (u64)i], n1 = n_strand[2 * (u64)i + 1], nn = n0 + n1, NE = P.num_extended;
    const VoteCand* c0 = A.out + m.pool_off[0];
    const VoteCand* c1 = A.out + m.pool_off[1];
    auto at = [&](int x) -> const VoteCand& { return x < n0 ? c0[x] : c1[x - n0]; };
    VoteCand* const top = sel + (u64)i * (u64)NE;
    if (nn <= kSelLds) {
        for (int a = lane; a < nn; a += 64) { u64 k1, k2; vote_keys(at(a), &k1, &k2); l_key[a] = make_ulonglong2(k1, k2); }
        __syncthreads();
        for (int a = lane; a < nn; a += 64) {
            const ulonglong2 ka = l_key[a];
            int rk = 0;
            for (int b = 0; b < nn; ++b) {
                const ulonglong2 kb = l_key[b];
                rk += (kb.x < ka.x || (kb.x == ka.x && (kb.y < ka.y || (kb.y == ka.y && b < a)))) ? 1 : 0;
            }
            if (rk < NE) top[rk] = at(a);
        }
    } else {
        for (int a = lane; a < nn; a += 64) {
            const VoteCand ca = at(a);
            int rk = 0;
            for (int b = 0; b < nn && rk < NE; ++b) {
                const VoteCand& cb = at(b);
                rk += (vote_before_dev(cb, ca) || (!vote_before_dev(ca, cb) && b < a)) ? 1 : 0;
            }
            if (rk < NE) top[rk] = ca;
        }
    }
    __syncthreads();
    const int mtop = nn < NE ? nn : NE;
    AsmPlanDev* const pl = plan + (u64)i * (u64)NE;
    int cnt = 0;
    for (int a0 = 0; a0 < mtop; a0 += 64) {
        const int a = a0 + lane;
        bool keep = false;
        VoteCand ca;
        if (a < mtop) {
            ca = top[a];
            keep = true;
            for (int b = 0; b < a && keep; ++b) keep = !(top[b].target_id == ca.target_id && top[b].chain == ca.chain);
        }
        const u64 km = __ballot(keep);
        if (keep) {
            AsmPlanDev e; e.sid = ca.target_id; e.sdir = ca.chain; e.qoff = -1; e.soff = -1; e.score = 0; e.ssize = ca.target_size;
            pl[cnt + popc64(km & below)] = e;
        }
        cnt += popc64(km);
    }
    if (lane == 0) nplan[i] = cnt;
}

// ------------------------------------------------------------------------------------------------------------------ range

// 32 elements of a strand, from strand position x on, in strand direction d (+1 / -1): element e = strand[x + d e].  g0 / len: the sequence in the
// volume; rev: the strand is the reverse complement (strand position x = 3 - base(g0 + len - 1 - x)).
NECAT_D u64 strand_word(const u64* bases, i64 g0, int len, int rev, int x, int d)
{
    return rev ? load32_dir(bases, g0 + len - 1 - x, -d, 1) : load32_dir(bases, g0 + x, d, 0);
}
NECAT_D u32 hash10_at(const u64* bases, i64 g0, int len, int rev, int x)
{
    return (u32)(rev2(strand_word(bases, g0, len, rev, x, +1)) >> (64 - 2 * kAsmRangeK));
}
NECAT_D u32 asm_mix(u32 h) { return h * 0x9E3779B1u; }

// entries of a sequence's k-mer list (build_kmif_list, find_mem.c:22-74): positions 0, w, 2 w, .. <= len - 10; the first entry's hash is 0
NECAT_D int kmif_count(int len, int window) { return len >= kAsmRangeK ? (len - kAsmRangeK) / window + 1 : 0; }
// where the one-byte occurrence counts of subject s, strand st start: 2 * (len / 6 + 1) bytes per subject never overlap the next one's
NECAT_D u64 occ_base(const u64* seq_off, u64 s, int st) { const u64 o = seq_off[s], len = seq_off[s + 1] - o; return 2 * (o / kAsmRangeW + s) + (u64)st * (len / kAsmRangeW + 1); }

// per (subject, strand): occurrences of every sampled 10-mer among the sampled ones.  tab: per wave `cap_max` u64 entries {key + 1, count}, all zero
// between items.
__global__ void __launch_bounds__(64)
k_asm_subj_occ(DevVolume ref, u32* __restrict__ tabs, u32 cap_max, u8* __restrict__ occ)
{
    const int lane = threadIdx.x;
    u32* const tab = tabs + (u64)blockIdx.x * 2 * cap_max;
    for (u64 item = blockIdx.x; item < 2 * ref.nseq; item += gridDim.x) {
        const u64 s = item >> 1;
        const int st = (int)(item & 1);
        const i64 g0 = (i64)ref.seq_off[s];
        const int len = (int)(ref.seq_off[s + 1] - ref.seq_off[s]);
        const int ns = kmif_count(len, kAsmRangeW);
        if (ns == 0) continue;
        u32 cap = 64; while (cap < 2u * (u32)ns) cap <<= 1;
        const u32 mask = cap - 1;
        for (int e = lane; e < ns; e += 64) {
            const u32 key = (e ? hash10_at(ref.bases, g0, len, st, e * kAsmRangeW) : 0u) + 1u;
            u32 p = asm_mix(key) & mask;
            for (;;) {
                const u32 old = atomicCAS(&tab[2 * p], 0u, key);
                if (old == 0u || old == key) break;
                p = (p + 1) & mask;
            }
            atomicAdd(&tab[2 * p + 1], 1u);
        }
        __syncthreads();
        u8* const dst = occ + occ_base(ref.seq_off, s, st);
        for (int e = lane; e < ns; e += 64) {
            const u32 key = (e ? hash10_at(ref.bases, g0, len, st, e * kAsmRangeW) : 0u) + 1u;
            u32 p = asm_mix(key) & mask;
            while (__hip_atomic_load(&tab[2 * p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != key) p = (p + 1) & mask;
            const u32 c = __hip_atomic_load(&tab[2 * p + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dst[e] = (u8)(c > 255u ? 255u : c);
        }
        __syncthreads();
        for (u32 p = (u32)lane; p < 2 * cap; p += 64) tab[p] = 0u;
        __syncthreads();
    }
}

// per read of the chunk: its 10-mer table.  keys / head: cap entries each (zero = empty), slot = mulhi(mix(key), cap), linear probing; next: one entry
// per position; a k-mer's positions hang off head[slot] as position + 1 (0 = end).
struct ReadIdxMeta { u64 tab_off, next_off; u32 cap, _pad; };          // tab_off: u32 entries (keys at tab_off, heads at tab_off + cap)
__global__ void __launch_bounds__(64)
k_asm_read_index(DevVolume reads, const u32* __restrict__ order, const ReadIdxMeta* __restrict__ rmeta, u32 n, u32* __restrict__ tabs, u32* __restrict__ nexts)
{
    const u32 i = blockIdx.x;
    if (i >= n) return;
    const int lane = threadIdx.x;
    const int r = (int)order[i];
    const i64 g0 = (i64)reads.seq_off[r];
    const int len = (int)(reads.seq_off[r + 1] - reads.seq_off[r]);
    const ReadIdxMeta m = rmeta[i];
    u32* const keys = tabs + m.tab_off; u32* const head = keys + m.cap; u32* const next = nexts + m.next_off;
    const int nk = kmif_count(len, 1);
    for (int e = lane; e < nk; e += 64) {
        const u32 key = (e ? hash10_at(reads.bases, g0, len, 0, e) : 0u) + 1u;
        u32 p = __umulhi(asm_mix(key), m.cap);
        for (;;) {
            const u32 old = atomicCAS(&keys[p], 0u, key);
            if (old == 0u || old == key) break;
            if (++p == m.cap) p = 0;
        }
        next[e] = atomicExch(&head[p], (u32)e + 1u);
    }
}

struct AsmSeed { i32 p, r, tl, tr; };          // subject position, read position, the extended match's [tl, tr) on the read
struct AsmMem { i32 len, q, r; };              // MaximalExactMatch (km_chain.h:11-15): match_size, query_offset (subject), reference_offset (read)
struct PairMeta { u32 read_i, slot; u64 seed_off; };      // read of the chunk, entry of its plan, first seed of the pair in the arena

// matches of the sampled subject 10-mer `e` with the read's 10-mers under find_kmer_match's limits (:92-133): their number, the first position in *first
NECAT_D int asm_probe(const u32* keys, const u32* head, const u32* next, u32 cap, u32 key, int occ_q, u32* first)
{
    if (occ_q > kAsmMaxOcc) return 0;
    u32 p = __umulhi(asm_mix(key), cap);
    for (;;) {
        const u32 kx = keys[p];
        if (kx == 0u) return 0;
        if (kx == key) break;
        if (++p == cap) p = 0;
    }
    int c = 0;
    const u32 h0 = head[p];
    for (u32 x = h0; x && c <= kAsmMaxOcc; x = next[x - 1]) ++c;
    if (c > kAsmMaxOcc || occ_q * c > kAsmMaxOcc) return 0;
    *first = h0;
    return c;
}

// matching elements of two strands from (xs, xr) on in direction d, at most `lim`
NECAT_D int asm_run(const u64* sb, i64 sg0, int slen, int srev, int xs, const u64* rb, i64 rg0, int rlen, int xr, int d, int lim)
{
    int nrun = 0;
    while (nrun < lim) {
        const u64 a = strand_word(sb, sg0, slen, srev, xs + d * nrun, d), b = strand_word(rb, rg0, rlen, 0, xr + d * nrun, d);
        const u64 x = a ^ b;
        int mrun = x ? ctz64(x) >> 1 : 32;
        const bool stop = mrun < 32;
        if (mrun > lim - nrun) mrun = lim - nrun;
        nrun += mrun;
        if (stop) break;
    }
    return nrun;
}

// EMIT = false: counts[w] = matches of pair w.  EMIT = true: the pair's matches (seeds arena, in subject-position order), extended; the surviving
// maximal exact matches of >= 15 bp sorted by (read offset, subject offset) into mems[seed_off ..], their number in nmem[w] (tmp: same size as mems).
template <bool EMIT>
__global__ void __launch_bounds__(64)
k_asm_seeds(DevVolume ref, DevVolume reads, const u32* __restrict__ order, const ReadIdxMeta* __restrict__ rmeta, const u32* __restrict__ tabs, const u32* __restrict__ nexts,
            const u8* __restrict__ occ, const AsmPlanDev* __restrict__ plan, int NE, const PairMeta* __restrict__ pairs, u32 npairs, u32* __restrict__ counts,
            AsmSeed* __restrict__ seeds, AsmMem* __restrict__ mems, AsmMem* __restrict__ tmp, u32* __restrict__ nmem)
{
    const u32 w = blockIdx.x;
    if (w >= npairs) return;
    const int lane = threadIdx.x;
    const u64 below = (1ULL << lane) - 1ULL;
    const PairMeta pm = pairs[w];
    const AsmPlanDev pe = plan[(u64)pm.read_i * (u64)NE + pm.slot];
    const int r = (int)order[pm.read_i];
    const i64 rg0 = (i64)reads.seq_off[r];
    const int rlen = (int)(reads.seq_off[r + 1] - reads.seq_off[r]);
    const i64 sg0 = (i64)ref.seq_off[pe.sid];
    const int slen = (int)(ref.seq_off[pe.sid + 1] - ref.seq_off[pe.sid]), srev = pe.sdir;
    const ReadIdxMeta rm = rmeta[pm.read_i];
    const u32* const keys = tabs + rm.tab_off; const u32* const head = keys + rm.cap; const u32* const next = nexts + rm.next_off;
    const u8* const oq = occ + occ_base(ref.seq_off, (u64)pe.sid, srev);
    const int ns = kmif_count(slen, kAsmRangeW), nkr = kmif_count(rlen, 1);
    AsmSeed* const sd = seeds + pm.seed_off;
    u32 total = 0;
    if (nkr > 0) for (int e0 = 0; e0 < ns; e0 += 64) {
        const int e = e0 + lane;
        int c = 0; u32 first = 0;
        if (e < ns) {
            const u32 key = (e ? hash10_at(ref.bases, sg0, slen, srev, e * kAsmRangeW) : 0u) + 1u;
            c = asm_probe(keys, head, next, rm.cap, key, (int)oq[e], &first);
        }
        if (EMIT) {
            const int inc = wave_prefix_sum(c);
            u32 at = total + (u32)(inc - c);
            for (u32 x = first; c > 0; x = next[x - 1], --c) {
                const int p = e * kAsmRangeW, rr = (int)x - 1;
                // extend_kmer_match (:173-220): to the left while both have bases, to the right from the ends of the 10-mers
                const int lmax = p < rr ? p : rr;
                const int lrun = lmax ? asm_run(ref.bases, sg0, slen, srev, p - 1, reads.bases, rg0, rlen, rr - 1, -1, lmax) : 0;
                const int rq = slen - (p + kAsmRangeK), rt = rlen - (rr + kAsmRangeK);
                const int rmax = rq < rt ? rq : rt;
                const int rrun = rmax > 0 ? asm_run(ref.bases, sg0, slen, srev, p + kAsmRangeK, reads.bases, rg0, rlen, rr + kAsmRangeK, +1, rmax) : 0;
                AsmSeed s; s.p = p; s.r = rr; s.tl = rr - lrun; s.tr = rr + kAsmRangeK + rrun;
                sd[at++] = s;
            }
            total += (u32)wave_read(inc, 63);
        } else total += (u32)c;
    }
    if (!EMIT) {
        for (int o = 32; o > 0; o >>= 1) total += (u32)__shfl_xor((int)total, o);
        if (lane == 0) counts[w] = total;
        return;
    }
    __syncthreads();
    // a match is dropped iff the nearest earlier match of its diagonal reaches beyond its start; only matches whose subject position lies within 10
    // of this match's left end can do that (a first entry's match carries 10 unverified bases)
    AsmMem* const tm = tmp + pm.seed_off;
    u32 nm = 0;
    for (u32 a0 = 0; a0 < total; a0 += 64) {
        const u32 a = a0 + (u32)lane;
        bool keep = false;
        AsmSeed s; s.p = s.r = s.tl = s.tr = 0;
        if (a < total) {
            s = sd[a];
            const int ql = s.p - (s.r - s.tl), diag = s.p - s.r;
            keep = true;
            for (i64 j = (i64)a - 1; j >= 0; --j) {
                const AsmSeed o = sd[j];
                if (o.p < ql - kAsmRangeK) break;
                if (o.p - o.r == diag && o.p < s.p) { keep = !(o.tr > s.r); break; }
            }
            if (s.tr - s.tl < kAsmMinMem) keep = false;
        }
        const u64 km = __ballot(keep);
        if (keep) { AsmMem mm; mm.len = s.tr - s.tl; mm.q = s.p - (s.r - s.tl); mm.r = s.tl; tm[nm + (u32)popc64(km & below)] = mm; }
        nm += (u32)popc64(km);
    }
    __syncthreads();
    // mem_before (find_mem.c:136-171): by read offset, then subject offset - a total order on the survivors
    AsmMem* const ms = mems + pm.seed_off;
    for (u32 a = (u32)lane; a < nm; a += 64) {
        const AsmMem x = tm[a];
        u32 rk = 0;
        for (u32 b = 0; b < nm; ++b) { const AsmMem y = tm[b]; rk += (y.r < x.r || (y.r == x.r && (y.q < x.q || (y.q == x.q && b < a)))) ? 1u : 0u; }
        ms[rk] = x;
    }
    if (lane == 0) nmem[w] = nm;
}

// ---- chain DP of one pair's sorted matches (scoring_mems, km_chain.c:141-205; parameters chain_data_new :5-23) and the best chain (mem_find_best_can)
constexpr int kMemMaxDist = 3000, kMemBw = 500, kMemMaxSkip = 25, kMemMinScore = 100;
constexpr int kLdsMems = 384;

NECAT_D bool mem_pair_score(const AsmMem& mi, const AsmMem& mj, int fj, int avg_cov, int* sc_out)
{
    if (mj.q + mj.len >= mi.q || mj.r + mj.len >= mi.r) return false;
    const int dr = mi.r - mj.r, dq = mi.q - mj.q;
    if (dr == 0 || dq <= 0) return false;
    if (dq > kMemMaxDist || dr > kMemMaxDist) return false;
    const int dd = dr > dq ? dr - dq : dq - dr;
    if (dd > kMemBw) return false;
    const int min_d = dq < dr ? dq : dr;
    int sc = min_d > mi.len ? mi.len : min_d;
    const int log_dd = dd ? 31 - __clz(dd) : 0;
    sc -= (int)((double)dd * .01 * (double)avg_cov) + (log_dd >> 1);
    *sc_out = sc + fj;
    return true;
}

__global__ void __launch_bounds__(64)
k_asm_chain(const PairMeta* __restrict__ pairs, u32 npairs, const AsmMem* __restrict__ mems, const u32* __restrict__ nmem, i32* __restrict__ chain_scratch,
            AsmPlanDev* __restrict__ plan, int NE)
{
    __shared__ AsmMem l_m[kLdsMems];
    __shared__ i32 l_f[kLdsMems], l_p[kLdsMems], l_t[kLdsMems], l_v[kLdsMems];
    const u32 w = blockIdx.x;
    if (w >= npairs) return;
    const int lane = threadIdx.x;
    const u64 below = (1ULL << lane) - 1ULL;
    const PairMeta pm = pairs[w];
    const int n = (int)nmem[w];
    if (n == 0) return;
    const AsmMem* gm = mems + pm.seed_off;
    const bool in_lds = n <= kLdsMems;
    const AsmMem* m = gm;
    // (volatile: beyond kLdsMems matches the arrays live in global memory, written and read by different lanes of this wave between barriers)
    volatile i32 *f, *p, *t, *v;
    if (in_lds) {
        for (int a = lane; a < n; a += 64) l_m[a] = gm[a];
        m = l_m; f = l_f; p = l_p; t = l_t; v = l_v;
    } else {
        i32* cs = chain_scratch + 4 * pm.seed_off;
        f = cs; p = cs + n; t = cs + 2 * (u64)n; v = cs + 3 * (u64)n;
    }
    int sum = 0;
    for (int a = lane; a < n; a += 64) { sum += gm[a].len; f[a] = 0; p[a] = -1; t[a] = 0; v[a] = 0; }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const int avg_cov = sum / n;
    __syncthreads();
    int st = 0;
    for (int i = 0; i < n; ++i) {
        const AsmMem mi = m[i];
        while (st < i && mi.r > m[st].r + kMemMaxDist) ++st;
        int max_f = mi.len, max_j = -1, n_skip = 0;
        for (int top = i - 1; top >= st; top -= 64) {
            const int j = top - lane;
            int sc = INT32_MIN;
            bool valid = false;
            if (j >= st) {
                valid = mem_pair_score(mi, m[j], f[j], avg_cov, &sc);
                if (valid) { const int pj = p[j]; if (pj >= 0) t[pj] = i; }
                else sc = INT32_MIN;
            }
            __syncthreads();
            const bool marked = valid && t[j] == i;
            const int incl = wave_prefix_max(sc);
            int before = wave_from_below(incl, max_f);
            if (before < max_f) before = max_f;
            const bool newmax = valid && sc > before;
            const u64 NM = __ballot(newmax), SK = __ballot(marked && !newmax);
            u64 live = ~0ULL;
            if (SK) {
                const u64 upto = below | (1ULL << lane);
                const int S = popc64(SK & upto) - popc64(NM & upto);
                int floor_ = wave_prefix_min(S);
                if (floor_ > -n_skip) floor_ = -n_skip;
                const int W = S - floor_;
                const u64 stop = __ballot(W > kMemMaxSkip);
                if (stop) live = (1ULL << ctz64(stop)) - 1ULL;
                else n_skip = wave_read(W, 63);
            } else {
                n_skip -= popc64(NM); if (n_skip < 0) n_skip = 0;
            }
            const u64 best = NM & live;
            if (best) { const int lb = 63 - __clzll((long long)best); max_f = wave_read(sc, lb); max_j = top - lb; }
            if (live != ~0ULL) break;
        }
        if (lane == 0) {
            f[i] = max_f; p[i] = max_j;
            v[i] = (max_j >= 0 && v[max_j] > max_f) ? v[max_j] : max_f;
        }
        __syncthreads();
    }
    // chain ends (no match has them as its predecessor) whose best prefix reaches the minimum score; the best by (peak score, peak index)
    for (int a = lane; a < n; a += 64) t[a] = 0;
    __syncthreads();
    for (int a = lane; a < n; a += 64) if (p[a] >= 0) t[p[a]] = 1;
    __syncthreads();
    i64 bestk = -1;
    for (int a = lane; a < n; a += 64) {
        if (t[a] == 0 && v[a] >= kMemMinScore) {
            int j = a;
            while (j >= 0 && f[j] < v[j]) j = p[j];
            if (j < 0) j = a;
            const i64 key = ((i64)f[j] << 32) | (i64)(u32)j;
            if (key > bestk) bestk = key;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { const i64 x = __shfl_xor(bestk, o); if (x > bestk) bestk = x; }
    if (bestk < 0) return;
    if (lane == 0) {
        const int jb = (int)(u32)(bestk & 0xffffffffLL), sc = (int)(bestk >> 32);
        int cnt = 0;
        for (int j = jb; j >= 0; j = p[j]) ++cnt;
        int jm = jb;
        for (int s = cnt - 1 - cnt / 2; s > 0; --s) jm = p[jm];
        const AsmMem mid = m[jm];
        AsmPlanDev* e = plan + (u64)pm.read_i * (u64)NE + pm.slot;
        // the roles come back exchanged (asm_pm_common.c:386-394): the chain's first sequence is the subject
        e->qoff = mid.r + mid.len / 2; e->soff = mid.q + mid.len / 2; e->score = sc;
    }
}

}  // namespace necat
