// knobs.h - the tuning / test knobs of a context.  Until round 5 these were process-wide globals re-read from the environment whenever ANY context was
// created: two contexts made with different environments raced on them.  Now every context holds the values its creator's environment had
// (necat_ctx::knobs, read once in necat_ctx_create), and an entry point of the C ABI makes its context's knobs the current ones for the length of the call
// (KnobScope, thread-local: one host thread per context, as include/necat_hip.h asks).  The g_* names the code uses are macros into the current set.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace necat {

struct Knobs {
    typedef uint32_t u32;
    // Lists with at most this many blocks use the cooperative DP kernel (k_myers_coop), longer ones the
    // lane-per-block kernel (k_myers).  With the band store filter the cooperative kernel is the faster one at
    // every size measured on MI355X (200 k blocks: 2.66 vs 2.80 ms; 50 k: 0.77 vs 1.38 ms), so the default is
    // "always"; NECAT_COOP_THRESHOLD=0 selects the lane-per-block kernel (tests compare the two).
    u32 coop_threshold;
    unsigned long long seed_budget;   // seeding scratch budget per chunk, in k-mer hits
    u32 batch_cap;       // candidates per extension batch (NECAT_BATCH)
    // NECAT_EXT_OVERLAP (default 1): a call of two or more batches runs them two at a time, side by side (two lanes: ExtLane, stage_extend.inl) - the other batch's
    // kernels fill the drain / ramp-up of every round's kernels and the ~ 15 latency-bound rounds a batch ends in.  0 = one batch after the other.
    // NECAT_EXT_OVERLAP_PCT (default 100): the next batch starts as soon as a lane is free; < 100: only when fewer than that per cent of the batch started last
    // still have a block to align (70: 293.7 against 277.5 ms per step at yeast size).
    // NECAT_EXT_OVERLAP_MIN (default 0 = never): a call of ONE batch of at least this many candidates is cut in two for the same effect, NECAT_EXT_OVERLAP_SPLIT per
    // cent (default 20: the longest chains) in the first.  E. coli size: 36.8 - 39.6 ms per step against 38.8 - 39.3 on one lane, depending on which of the process's
    // streams the runtime has put on one hardware queue (tools/r05/run19, run22, run24): not a reliable gain, so not the default - which also keeps the bench line's
    // roofline (priced on its kernels' event durations) free of launches that share the chip.
    u32 ext_overlap_order;       // NECAT_EXT_ORDER=0: several batches take the candidates as they come instead of longest expected chain first (A/B)
    u32 ext_overlap, ext_overlap_min, ext_overlap_pct, ext_overlap_split;      
    u32 single_pass;     // lists up to this many blocks use the single-pass DP kernel (NECAT_SINGLE_PASS; 0 = never)
    int index_lds;       // LDS-slice index passes (NECAT_INDEX_LDS=0: global-atomic bucket passes)
    int split_threads;   // NECAT_SPLIT_THREADS (512, or 256 = until round 5): threads of a workgroup of the index build's split kernels (k_split_bases, k_split_recs, k_subpart), each on a 4096-record tile
    int seed_wave;       // wave-per-strand seed collection (NECAT_SEED_WAVE=0: the lane-per-strand kernel)
    int seed_kst;        // NECAT_SEED_KST=0: k_seed_collect_wave looks the table up again instead of reading the words k_seed_hits kept (A/B tests)
    int trace;           // NECAT_TRACE: 1 = extension rounds, 2 = host stages
    int coop_filter;     // NECAT_COOP_FILTER=0: the cooperative kernel stores every word (A/B tests)
    int sort_b;          // NECAT_SORT_B=0 disables the size sort of list B
    int cns_spec_extra, cns_spec_cover;   // NECAT_CNS_SPEC_EXTRA / NECAT_CNS_SPEC: speculation width of the consensus loop
    int fast;            // NECAT_FAST=0: the list-A DP kernel never takes its full-block fast path (A/B measurements); 2: fast path without band stores (profiling only, results invalid)
    int fast16;          // NECAT_FAST16=1: list A's big rounds through k_myers_a16 (16 full blocks per workgroup: SHW 8 lanes, NW 4 lanes per block)
    size_t band_pool;    // NECAT_BAND_POOL_MB (default 16384): cap of one band-record pool; a bigger list runs in several DP + walk launches (0 = no cap)
    int walk;            // NECAT_WALK=0: k_traceback runs the reference formulation of the walk (A/B measurements)
    u32 tail_fused;      // NECAT_TAIL_FUSED (default 512 = one workgroup per block at 2 per CU; 0 = off): lists of at most this many blocks run as ONE launch per round with the band in LDS (ext_tail.h)
    u32 rcwalk;          // NECAT_RCWALK (default 512 = every list the one-launch tail kernel does not take; 0 = off): list-A rounds of more than this many blocks run through k_myers_ck / k_myers_ckg + k_rcwalk2 (ext_rcwalk.h: no NW pass, no band records, the walk recomputes its cells)
    size_t rc_pool;      // NECAT_RC_POOL_MB (default 8192 = 1.6 M list-A blocks per launch; a 0.6 Gbp volume: 182 -> 174 ms per pass against 2048): cap of the checkpoint buffer of those rounds; a longer list goes through it in several launches
    u32 asm_rc;          // NECAT_ASM_RC (default 1): the 2048-bp block aligner of oc2asmpm through k_myers_ckg + k_rcwalk2 (no NW pass, no band records); 0 = two-pass kernel + band + wave walk
    u32 rc_listb;        // NECAT_RC_LISTB (default 1, needs NECAT_RC_CARRY): list B (blocks up to 794 x 794) through k_myers_ckg + k_rcwalk2 too; 0 = two-pass kernel + band pool + walk
    u32 rc_ragged;       // NECAT_RC_RAGGED (default 1, needs NECAT_RC_CARRY): the ragged blocks of those rounds through k_myers_ckg + k_rcwalk2 as well (0: two-pass kernel + lane walk on a stream of their own)
    u32 ck_lds;          // NECAT_CK_LDS (bytes, default 0): dynamic LDS claimed by every workgroup (one wave) of k_myers_ck - caps how many of its waves a CU holds (160 KB / (1 KB + this)), leaving wave slots to the chains of the other streams (A/B measurements)
    u32 frag_fuse;       // NECAT_FRAG_FUSE (default 1; needs the merged big-round path): list A's checkpoint pass cuts its blocks' fragments out of the volumes itself (k_myers_ck flag bit 22) and k_round_ctl does the round's bookkeeping; 0 = k_ext_frag in a launch of its own before every pass, as until round 5
    u32 rc_merge;        // NECAT_RC_MERGE (default 1, needs NECAT_RC_RAGGED): the ragged list-A blocks of a big round through k_myers_ck's ragged fast path and the full blocks' walk launch; 0 = k_myers_ckg + a walk launch of their own on stream d
    u32 rc_prio;         // NECAT_RC_PRIO (bits; default 1: 41.6 -> 41.0 ms per step; 2 costs 0.5 ms, 4 nothing): waves that raise their issue priority (s_setprio 3) - 1: list A's walk (k_rcwalk2w: every wave; k_rcwalk3: its walking wave), 2: list A's checkpoint pass, 4: list B's walk, 8 / 16: only the WALKING wave of list A's / list B's walk, for the length of its walk (kernel opts bit 16)
    u32 rc_pipe, rc_pipe_min;    // NECAT_RC_PIPE (default 1 = off: 2 - 4 pieces cost 1.8 - 2.3 ms per step, tools/r04/run28.sh, run29.sh) / NECAT_RC_PIPE_MIN (default 49152 blocks): list A of a big round in pieces, walk of piece i beside the pass of piece i + 1
    u32 ext_lanes;       // NECAT_EXT_LANES (1 .. 4, default 2): lanes a call of several batches runs its batches on side by side (stage_extend.inl; 1 = NECAT_EXT_OVERLAP=0)
    u32 ckr_fast;        // NECAT_CKR_FAST (default 1): list B's checkpoint pass (fast_shw_ckr in k_myers_ckf) runs the windows in which every lane of the wave is inside its block unrolled and without a per-step lane mask; 0 = every window rolled, as until round 5
    u32 ck_post;         // NECAT_CK_POST (default 1): k_myers_ck finds the bottom row's minimum after the pass, from word 7's deltas, and unrolls its windows (fast_shw8_ckp); 0 = tracked inside the pass
    u32 rc_fastb;        // NECAT_RC_FASTB (default 1): list B's checkpoint pass through k_myers_ckf (32-bit halves, bitop3, DPP carries); 0 = the general pass k_myers_ckg
    u32 rc_dbg;          // NECAT_RC_DBG (timing only): 2 = k_rcwalk2w walks every segment twice (once into a sink), 4 = recomputes every segment twice
    u32 rc_prefetch;     // NECAT_RC_PREFETCH (default 0: measured 0.4 ms per step SLOWER, profiles/NOTES_r04.md 3): k_rcwalk2w loads the next segment's checkpoints / deltas / planes a segment ahead
    u32 rc_ww;           // NECAT_RC_WW (default 1): which recompute walk runs. 1 = k_rcwalk2w (64-row records, one LDS read per walk step), except that list-A launches of at least NECAT_RC3_MIN blocks go through k_rcwalk3; 2 = k_rcwalk3 everywhere (ext_rcwalk3.h: a workgroup of TWO waves recomputes 64 blocks - two lanes per block, both words of the pair per lane - into 32-DIAGONAL records, one of the two then walks the blocks column by column: faster alone, slower in the bench's small launches, profiles/NOTES_r05.md 1); 0 = k_rcwalk2 (every lane of a quad walks its block: cross-check build only - the product library refuses it when the context is created, necat_ctx_create prints why)
    u32 rc3_band;          // NECAT_RC3_BAND (32 or 16): diagonals per record of k_rcwalk3 (16: half the LDS per block in flight, 7 waves per SIMD instead of 4.5, a few per cent of the segments redone)
    u32 rc3_min;         // NECAT_RC3_MIN (blocks, default 160000; 4294967295 = never): with NECAT_RC_WW=1, list-A launches of at least this many blocks go through k_rcwalk3 (throughput form: fewer instructions per block, longer chain per segment) instead of k_rcwalk2w
    u32 rc_carry;        // NECAT_RC_CARRY (default 1): the recompute walk on an exact two-word window (k_myers_ck<CARRY> keeps the words' horizontal deltas, k_rcwalk2); 0 = the 4-word band window (k_rcwalk4)
    int rc_maxdist;      // NECAT_RC_MAXDIST (default and maximum kRcMaxDist = 160): full blocks of a larger distance take the old kernels (tests lower it)
    u32 walk_wave;       // NECAT_WALK_WAVE (default 12288; 0 = off): lists of at most this many blocks are walked by one WAVE per block through an LDS window (k_walk_wave, ext_tail.h)
    int asm_lane;        // NECAT_ASM_LANE=1: necat_asm_align_batch through the lane-per-alignment kernel (k_asm_align), the second implementation
    int dbg;             // NECAT_DBG: profiling-only variants of the lane-per-block DP kernel (1 = no band stores, 2 = no NW pass)
};

extern thread_local const Knobs* tl_knobs;      // the knobs of the context whose call is running on this thread (necat_hip.hip)

}  // namespace necat

#define g_coop_threshold (necat::tl_knobs->coop_threshold)
#define g_seed_budget (necat::tl_knobs->seed_budget)
#define g_batch_cap (necat::tl_knobs->batch_cap)
#define g_ext_overlap (necat::tl_knobs->ext_overlap)
#define g_ext_overlap_min (necat::tl_knobs->ext_overlap_min)
#define g_ext_overlap_pct (necat::tl_knobs->ext_overlap_pct)
#define g_ext_overlap_split (necat::tl_knobs->ext_overlap_split)
#define g_ext_overlap_order (necat::tl_knobs->ext_overlap_order)
#define g_single_pass (necat::tl_knobs->single_pass)
#define g_index_lds (necat::tl_knobs->index_lds)
#define g_split_threads (necat::tl_knobs->split_threads)
#define g_seed_wave (necat::tl_knobs->seed_wave)
#define g_seed_kst (necat::tl_knobs->seed_kst)
#define g_trace (necat::tl_knobs->trace)
#define g_coop_filter (necat::tl_knobs->coop_filter)
#define g_sort_b (necat::tl_knobs->sort_b)
#define g_cns_spec_extra (necat::tl_knobs->cns_spec_extra)
#define g_cns_spec_cover (necat::tl_knobs->cns_spec_cover)
#define g_fast (necat::tl_knobs->fast)
#define g_fast16 (necat::tl_knobs->fast16)
#define g_band_pool (necat::tl_knobs->band_pool)
#define g_walk (necat::tl_knobs->walk)
#define g_tail_fused (necat::tl_knobs->tail_fused)
#define g_rcwalk (necat::tl_knobs->rcwalk)
#define g_rc_pool (necat::tl_knobs->rc_pool)
#define g_asm_rc (necat::tl_knobs->asm_rc)
#define g_rc_listb (necat::tl_knobs->rc_listb)
#define g_rc_ragged (necat::tl_knobs->rc_ragged)
#define g_ck_lds (necat::tl_knobs->ck_lds)
#define g_rc_merge (necat::tl_knobs->rc_merge)
#define g_frag_fuse (necat::tl_knobs->frag_fuse)
#define g_rc_prio (necat::tl_knobs->rc_prio)
#define g_rc_pipe (necat::tl_knobs->rc_pipe)
#define g_rc_pipe_min (necat::tl_knobs->rc_pipe_min)
#define g_ck_post (necat::tl_knobs->ck_post)
#define g_ckr_fast (necat::tl_knobs->ckr_fast)
#define g_ext_lanes (necat::tl_knobs->ext_lanes)
#define g_rc_fastb (necat::tl_knobs->rc_fastb)
#define g_rc_dbg (necat::tl_knobs->rc_dbg)
#define g_rc_prefetch (necat::tl_knobs->rc_prefetch)
#define g_rc_ww (necat::tl_knobs->rc_ww)
#define g_rc3_min (necat::tl_knobs->rc3_min)
#define g_rc3_band (necat::tl_knobs->rc3_band)
#define g_rc_carry (necat::tl_knobs->rc_carry)
#define g_rc_maxdist (necat::tl_knobs->rc_maxdist)
#define g_walk_wave (necat::tl_knobs->walk_wave)
#define g_asm_lane (necat::tl_knobs->asm_lane)
#define g_dbg (necat::tl_knobs->dbg)
