// asm_coop.h - the block aligner of oc2asmpm (blockwise_edlib_align, asm_pm/blockwise_edlib.c:1205-1371 = onc_align with 2048-bp blocks, tail
// match length 8; called by hbn_map_extend, asm_pm/hbn_align.c:282-326) through the extension stage's own machinery at a bigger geometry:
//
//   every (read, subject strand, anchor) triple is an ExtTask that advances one block per ROUND (ext_plan<2048>); a round is (necat_asm_align_batch, necat_hip.hip)
//   k_ext_frag<32 | 44, ..>      fragments of every scheduled block from the 2-bit volumes (the subject on either strand)
//   k_myers_ckg<32 | 44, ..>     ONE DP pass (ext_rcwalk.h): lane b of the wave owns 64-row word b of a block (32 lanes per block up to 2048 x 2048, a whole wave for
//                                the longer last blocks - (2048 + 99) x 1.3 = 2791 bases, 44 words), anti-diagonal wavefront, carries by DPP; it keeps every word's
//                                (Pv, Mv) after every 16th column and the words' horizontal deltas - no NW pass, no band records
//   k_rcwalk2w<32 | 44, ..>      the walk, which recomputes the two words it stands on from those checkpoints (up > left > diagonal)
//   k_traceback<.., WALK = 5>    the tail trimmed at the last run of 8 matches, the kept columns appended to the task's 2-bit column stream, the next block
//                                planned and appended to the next round's list
//   and after the last round k_ext_alignment / k_ext_strings give every anchor's coordinates, identity and alignment columns - exactly the outputs
//   of necat_onc_align_batch.
// Two lists per round as in the 512-bp stage (A: blocks up to 2048 x 2048; B: the longer last blocks), rounds synchronous on the host - a corrected read is
// 3 - 5 blocks long, so a call is a handful of rounds.  NECAT_ASM_RC=0 runs the round-3 form instead - k_myers_coop (below: SHW pass + NW pass with band
// records) and the band walk - and NECAT_ASM_LANE=1 the lane-per-alignment kernel of round 2 (k_asm_align, asm_kernels.h): the two older implementations
// the tests compare with.
#pragma once
#include "asm_kernels.h"
#include "ext_kernels.h"

namespace necat {

// two lists as in the 512-bp stage: A = blocks of at most 2048 x 2048 (32 words: 32 lanes per block, two blocks per wave), B = the
// longer last blocks (up to 2791 x 2791, 44 words: one block per wave)
constexpr int kAsmWordsA = kAsmBlock / 64, kAsmTWordsA = kAsmBlock / 32, kAsmOpsA = 2 * kAsmBlock + 16;
constexpr int kAsmFragWordsA = 2 * kAsmWordsA + kAsmTWordsA;
constexpr size_t kAsmSlabA = (size_t)kAsmBlock * kAsmWordsA * 64 * sizeof(BandRec);          // 67 MB per 64 items (sparsely written)
constexpr int kAsmTWords = (kAsmCols + 31) / 32;                  // 88 target words of 32 columns
constexpr int kAsmFragWords = 2 * kAsmWords + kAsmTWords;         // u64 per item in the fragment buffer
constexpr int kAsmMaxOps = 2 * kAsmCols + 16;                     // ops of one block alignment (<= qn + tn)
constexpr size_t kAsmSlab = (size_t)((kAsmCols + 7) & ~7) * kAsmWords * 64 * sizeof(BandRec);     // band records of 64 items (126 MB, sparsely written)

// tasks from anchors: the query is the forward read, the subject on strand sdir (ExtTask::sdir; ext_frag_geom reads it backwards and
// complemented); first block planned and appended to the round-0 list
__global__ void __launch_bounds__(256)
k_asm_init(const AsmAnchor* __restrict__ anchors, u32 n, const u64* __restrict__ reads_off, const u64* __restrict__ ref_off, ExtTask* __restrict__ tasks,
           ExtLists L, const u64* __restrict__ ops_base)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    ExtTask t;
    bool go = false;
    if (i < n) {
        const AsmAnchor a = anchors[i];
        ext_init(t, (i32)i, 0, (i64)reads_off[a.q], (i32)(reads_off[a.q + 1] - reads_off[a.q]), (i64)ref_off[a.s], (i32)(ref_off[a.s + 1] - ref_off[a.s]), a.qoff, a.soff);
        t.sdir = a.sdir;
        t.ops_base = ops_base[i];
        go = ext_plan<kAsmBlock>(t);
        tasks[i] = t;
    }
    ext_append_block_wg<kAsmBlock, false, 4>(t, i, go, L);
}

}  // namespace necat
