// ext_tail.h - the late rounds of an extension in ONE launch per round, the band in LDS.
//
// Once only the longest chains are alive a round holds a few hundred blocks (list A <= NECAT_TAIL_FUSED items) and lasts exactly
// as long as ONE block alignment: fragments -> DP -> walk, a chain of three launches whose every link is latency bound (profiles/
// r02_round_timeline.txt: frag 5 us + single-pass DP 0.12 ms + walk 0.23 ms + two kernel boundaries, a dozen such rounds per pass,
// three passes in the consensus loop).  The walk is the long link: one dependent 16-byte global load per step (~ 0.45 us each),
// 520 steps.  Here one workgroup (one wave) owns one block alignment from the volumes to the candidate's next block:
//   * the fragments go straight from the 2-bit volumes into registers / LDS (no fragment buffer round trip);
//   * the single-pass DP (lane b = 64-row word b, anti-diagonal wavefront, DPP carries - myers_coop_wave's SINGLE path) writes
//     its band records to LDS: 512 columns x 8 words x 16 bytes = 64 KB of the CU's 160 KB;
//   * the walk (walk_block, dp_core.h) reads them back from LDS - ~ 100 cycles per dependent read instead of ~ 1000 - and leaves
//     its ops in LDS too;
//   * tail trimming, the candidate's counters, its kept alignment columns (when the caller keeps them) and the plan of the next
//     block follow in the same kernel (what k_traceback does after its walk).
// Results are those of the three-kernel chain bit for bit (same cores: advance_dev, walk_block / ext_finish_block / ext_plan).
// List B too: a last block has at most 611 bases on ONE side (get_next_sequence_block, oc_aligner.c:127-131: the short side is
// what is left, < 612, the other one at most 1.3 x that), so words x columns <= 10 x 794 < 13 x 611 = 7943 records = 127 KB with the
// band laid out [column][word of THIS block] - one workgroup per CU.  (CAP = that bound; a block beyond it raises the error flag.)
#pragma once

namespace necat {

constexpr int kTailThreads = 64;
constexpr int kTailCapB = 13 * 611 + 1;       // band records of the biggest list-B block: 13 words x 611 columns (10 x 794 is smaller)

struct LdsBand {          // Mat of walk_block: band[c * nblk + b]
    const ulonglong2* band; int nblk;
    NECAT_D void rec(int c, int b, u64& A, u64& B) const { const ulonglong2 v = band[c * nblk + b]; A = v.x; B = v.y; }
};
struct LdsOpsReader { const u8* ops; NECAT_D int operator()(int j) const { return ops[j]; } };
struct LdsOpsSink {       // Sink of walk_block: op i at ops[i] (every lane of the wave runs the same walk; lane 0 writes)
    u8* ops; int cap; int overflow; bool store, writer;
    NECAT_D bool storing() const { return store; }
    NECAT_D void put(int i, int op) { if (i < cap) { if (writer) ops[i] = (u8)op; } else overflow = 1; }
};

// Band records of ONE block read from its slab in global memory through a window staged in LDS by the whole wave: every lane runs
// the same walk (same values, uniform control flow), and when the walk asks for a record outside the window all 64 lanes fetch
// the 64 columns (c, c - 1, .., c - 63) of the word asked for and of the word above it - one load instruction per word, eight
// 128-byte lines each ([column / 8][word][lane][column % 8], rec_pos) - so the walk waits for memory once per <= 64 columns instead
// of once per step (the lane-per-block walk of k_traceback: one dependent 16-byte load per step, ~ 0.45 - 1 us each when few
// blocks are in flight).
template <int NW>
struct WinBand {
    const ulonglong2* slab; ulonglong2* win;      // win[2][64] in LDS
    int il, lane, wb, chi;
    NECAT_D void init() { wb = -1000; chi = -1000; }
    NECAT_D void rec(int c, int b, u64& A, u64& B)
    {
        const bool hit = (b == wb || b == wb - 1) && c <= chi && c > chi - 64;
        if (!hit) {                 // wave-uniform
            __syncthreads();        // (one wave per workgroup) nobody still reads the old window
            wb = b; chi = c;
            const int col = c - lane;
            if (col >= 0) {
                win[lane] = slab[rec_pos<NW>(col, b, il)];
                if (b > 0) win[64 + lane] = slab[rec_pos<NW>(col, b - 1, il)];
            }
            __syncthreads();
        }
        const ulonglong2 v = win[(b == wb ? 0 : 64) + (chi - c)];
        A = v.x; B = v.y;
    }
};

// What follows a block's DP, by ONE wave that owns the block: the walk (every lane runs it, see WinBand / LdsBand), then on lane 0
// the tail trimming + the candidate's counters (ext_finish_block), by all lanes the kept alignment columns into the task's 2-bit
// stream (when the caller keeps them), on lane 0 the plan of the next block, and the append to the next round's lists.
// ops: MAXOPS bytes of LDS.  The tail of k_traceback, for a wave instead of a lane.
template <int MAXOPS, int BLOCK, class Mat, class Same>
NECAT_D void wave_after_dp(const int lane, const BlockItem& it, const int dist, const int endc, Mat& mat, Same& same, u8* ops,
                           ExtTask* __restrict__ tasks, const int tail_match_len, int* __restrict__ err_flag, const ExtLists& next)
{
    ExtTask t;
    ExtKept kept; kept.at = 0; kept.cols = 0; kept.exact = 0;
    int nops = 0, stream_at = 0, done = 0, found = 0;
    bool go = false;
    if (lane == 0) { t = tasks[it.task]; done = ext_block_done(t, dist, endc); found = t.found; }
    done = __shfl(done, 0); found = __shfl(found, 0);
    TailScan ts;
    tail_init(ts, !done ? kOcaMatCnt : tail_match_len);
    LdsOpsSink sk; sk.ops = ops; sk.cap = MAXOPS; sk.overflow = 0; sk.writer = lane == 0;
    sk.store = !found || next.task_ops != nullptr;       // the op list is only replayed until the stream's first run of 8 matches - unless the columns are kept
    if (dist >= 0) {
        walk_block(it.qn, endc + 1, mat, sk, ts);
        if (sk.overflow && lane == 0) atomicExch(err_flag, 20);
    }
    __syncthreads();              // lane 0's ops are in LDS
    if (lane == 0) {
        LdsOpsReader rd; rd.ops = ops;
        stream_at = t.phase == 1 ? t.s_lto : 0;
        kept = ext_finish_block(t, dist, endc, done, ts, rd, same);
        nops = ts.n;
    }
    if (next.task_ops) {
        // the kept columns join the task's stream, 2 bits per column: forward column f of the block is op nops - 1 - f (k_traceback)
        const u64 ops_base = __shfl(lane == 0 ? t.ops_base : 0ULL, 0);
        const int at = __shfl(stream_at + kept.at, 0), ncol = __shfl(kept.cols, 0), exact = __shfl(kept.exact, 0), n = __shfl(nops, 0);
        u64* reg = reinterpret_cast<u64*>(next.task_ops + ops_base);
        // words of the stream this block touches: [at, at + ncol) columns; the first one may hold earlier columns (kept), the last
        // one's upper bits are zero (the next block ORs into them)
        const int w0 = at >> 5, w1 = (at + ncol + 31) >> 5;
        for (int w = w0 + lane; w < w1; w += 64) {
            u64 acc = 0;
            const int c_lo = w * 32 > at ? w * 32 : at, c_hi = (w + 1) * 32 < at + ncol ? (w + 1) * 32 : at + ncol;
            if (w * 32 < at) acc = reg[w] & ((1ULL << ((at & 31) * 2)) - 1);
            if (!exact) for (int col = c_lo; col < c_hi; ++col) acc |= (u64)ops[n - 1 - (col - at)] << ((col & 31) * 2);
            reg[w] = acc;
        }
    }
    if (lane == 0) {
        go = ext_plan<BLOCK>(t);
        tasks[it.task] = t;
    }
    ext_append_block<BLOCK>(t, (u32)it.task, go, next);
}

// The walk of a SMALL list with one wave per block (WinBand): the DP kernels ran as usual (band records in the list's slabs), this
// replaces k_traceback for lists where a lane-per-block walk would leave the chip empty and pay a memory round trip per step.
template <int NW, int TW, int MAXOPS, int BLOCK = kOcaBlockSize>
__global__ void __launch_bounds__(64)
k_walk_wave(const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, const u64* __restrict__ frag, const char* __restrict__ slabs, size_t slab_bytes,
            const BlockResult* __restrict__ results, ExtTask* __restrict__ tasks, int tail_match_len, int* __restrict__ err_flag, ExtLists next, u32 item_base)
{
    constexpr int FW = 2 * NW + TW;
    __shared__ ulonglong2 win[128];
    __shared__ u8 ops[MAXOPS];
    const int lane = (int)threadIdx.x;
    const ListView lv = list_view(n_host, n_dev, capA);
    const u64 item = (u64)item_base + blockIdx.x;
    BlockItem it;
    if (!list_item(lv, items, item, it)) return;          // uniform: the workgroup is one work item
    const BlockResult br = results[item];
    if (br.err && lane == 0) atomicExch(err_flag, 10 + br.err);
    const u64 grp = item >> 6;
    WinBand<NW> mat; mat.slab = slab_records(const_cast<char*>(slabs) + (size_t)grp * slab_bytes); mat.win = win; mat.il = (int)(item & 63); mat.lane = lane; mat.init();
    SameReader<NW> same; same.fr = frag + grp * FW * 64 + (item & 63);
    wave_after_dp<MAXOPS, BLOCK>(lane, it, br.dist, br.endc, mat, same, ops, tasks, tail_match_len, err_flag, next);
}
template <int NW>
struct LdsSame {          // query fragment element i == target fragment element i ?  (fr: [~lo planes NW][~hi planes NW][target 2-bit words])
    const u64* fr;
    NECAT_D bool operator()(int i) const
    {
        const u64 nlo = fr[i >> 6], nhi = fr[NW + (i >> 6)];
        const int q = (int)((~nlo >> (i & 63)) & 1) | ((int)((~nhi >> (i & 63)) & 1) << 1);
        return q == (int)((fr[2 * NW + (i >> 5)] >> ((i & 31) * 2)) & 3);
    }
};

// the round's bookkeeping as a launch of its own (k_ext_frag does it in the three-kernel chain): list B's chain of the round waits
// for this, not for the whole fused list-A kernel
__global__ void __launch_bounds__(64)
k_round_ctl(RoundCtl ctl)
{
    if (threadIdx.x == 0 && ctl.pub) {
        const u32 a = ctl.count[0] + ctl.count[2], b = ctl.count[1];
        ctl.zero[0] = 0u; ctl.zero[1] = 0u; ctl.zero[2] = 0u;
        volatile RoundPub* p = ctl.pub;
        p->nA = a; p->nB = b;
        __threadfence_system();
        p->seq = ctl.seq;
        __threadfence_system();
    }
}

template <int NW, int TW, int CAP, int MAXOPS>
__global__ void __launch_bounds__(kTailThreads)
k_tail_fused(DevVolume reads, DevVolume ref, const BlockItem* __restrict__ items, u32 n_host, const u32* __restrict__ n_dev, u32 capA, double error,
             ExtTask* __restrict__ tasks, int tail_match_len, int* __restrict__ err_flag, ExtLists next, unsigned long long* __restrict__ stats)
{
    __shared__ ulonglong2 band[CAP];
    __shared__ u64 fr[2 * NW + TW];
    __shared__ u64 tpl[TW];
    __shared__ u8 ops[MAXOPS];
    __shared__ int res[2];
    const int lane = (int)threadIdx.x;
    const ListView lv = list_view(n_host, n_dev, capA);
    BlockItem it;
    if (!list_item(lv, items, (u64)blockIdx.x, it)) return;          // uniform: the workgroup is one work item
    const int qn = it.qn, tn = it.tn;
    const int nblk = (qn + 63) >> 6, W = nblk * 64 - qn;
    if (nblk * tn > CAP) { if (lane == 0) atomicExch(err_flag, 30); return; }
    // ---- fragments: lane b < NW the query word b (two complemented bit-planes), lanes NW .. NW + TW - 1 the target words
    u64 nlo = 0, nhi = 0;
    if (lane < NW) {
        if (lane * 64 < qn) { u64 lo, hi; load64_planes(reads.bases, it.g.q_base, it.g.q_dir, it.g.q_comp, lane * 64, &lo, &hi); nlo = ~lo; nhi = ~hi; }
        fr[lane] = nlo; fr[NW + lane] = nhi;
    } else if (lane < NW + TW) {
        const int tw = lane - NW;
        const u64 x = tw * 32 < tn ? load32_dir(ref.bases, it.g.t_base + (i64)it.g.t_dir * (tw * 32), it.g.t_dir, it.g.t_comp) : 0ULL;
        fr[2 * NW + tw] = x;
        tpl[tw] = even_bits(x) | (even_bits(x >> 1) << 32);
    }
    __syncthreads();
    // ---- single-pass DP (edlib_ex.c:108-223 with every word computed; the SINGLE path of myers_coop_wave), records to LDS
    if (lane < NW) {
        const int b = lane;
        const bool have = b < nblk, is_last = have && b == nblk - 1;
        const u64 pad = (is_last && W > 0) ? (~0ULL << ((64 - W) & 63)) : 0ULL;
        const u32 nlo_l = (u32)nlo, nlo_h = (u32)(nlo >> 32), nhi_l = (u32)nhi, nhi_h = (u32)(nhi >> 32);
        const u32 pad_l = (u32)pad, pad_h = (u32)(pad >> 32);
        u64 tcur = 0;
        auto eq_of = [&](int c) -> u64 {
            const u32 ma = (u32)__builtin_amdgcn_sbfe((int)(u32)tcur, (u32)c & 31u, 1u);
            const u32 mb = (u32)__builtin_amdgcn_sbfe((int)(u32)(tcur >> 32), (u32)c & 31u, 1u);
            const u32 el = ((nlo_l ^ ma) & (nhi_l ^ mb)) | pad_l, eh = ((nlo_h ^ ma) & (nhi_h ^ mb)) | pad_h;
            return ((u64)eh << 32) | el;
        };
        const int steps = tn + nblk - 1;
        int k = (int)((double)(qn < tn ? qn : tn) * error * 1.1);
        u64 P = ~0ULL, M = 0ULL;
        int S = (b + 1) * 64, best = -1, end0 = -1, hout = 1;
        for (int s = 0; s < steps; ++s) {
            const int c = s - b;
            int hin = dpp_from_lane_below(hout);
            if (b == 0) hin = 1;
            if (have && (u32)c < (u32)tn) {
                if ((c & 31) == 0) tcur = tpl[c >> 5];
                const u64 eq = eq_of(c);
                u64 rA, rB;
                hout = advance_dev<true>(P, M, eq, hin, rA, rB);
                S += hout;
                band[c * nblk + b] = make_ulonglong2(rA, rB);
                if (is_last && S <= k && (best == -1 || S <= best)) {
                    if (S != best) { best = S; k = best; end0 = c - W; }
                }
            }
        }
        if (is_last) {
            if (W > 0) {          // edlib_ex.c:205-219
                int score = S;
                for (int i = 0; i < W; ++i) {
                    if (P & (kHighBit >> i)) --score;
                    if (M & (kHighBit >> i)) ++score;
                    if (score <= k && (best == -1 || score <= best)) {
                        if (score != best) { k = best = score; end0 = tn - W + i; }
                    }
                }
            }
            res[0] = best; res[1] = end0;
        }
    }
    __syncthreads();
    // ---- walk, trimming, the candidate's counters, its kept columns, its next block
    const int dist = res[0], endc = res[1];
    if (lane == 0) { stat_add(stats, 0, (unsigned long long)(nblk * tn)); stat_add(stats, 1, (unsigned long long)(qn + tn)); }
    LdsBand mr; mr.band = band; mr.nblk = nblk;
    LdsSame<NW> same; same.fr = fr;
    wave_after_dp<MAXOPS, kOcaBlockSize>(lane, it, dist, endc, mr, same, ops, tasks, tail_match_len, err_flag, next);
}

}  // namespace necat
