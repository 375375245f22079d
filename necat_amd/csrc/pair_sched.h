// pair_sched.h - static schedule of the overlap stage's volume-pair jobs over the GPUs of one node (both granularities
// combined; BASELINE configs[3] / [4]).  Host code, no HIP.
//
// What there is to schedule: volume v of a project is the reference of one oc2pmov job that maps query volumes v, v + 1, .., V - 1
// against it (pm_worker.c:372-390; necat.pl:190-202 sends those jobs to grid nodes whole).  That is V (V + 1) / 2 (reference, query)
// pairs of very unequal weight: volume 0 has V query volumes and volume V - 1 one, the last volume of a project is a remainder
// (makedb/main.c:29 closes a volume at 2 Gbp), and the self pair maps every read only against the reads before it
// (word_finder.c:121-127): half a pair.  Whole reference volumes dealt to ranks (what round 2 did) leave the ranks of the late
// volumes idle: Drosophila's three volumes on four GPUs finish 3 : 2 : 1 : 0.
//
// Here the pairs are laid end to end on one cost line, in job order (v ascending, then i ascending), and rank g takes the stretch
// [g C / G, (g + 1) C / G) of it.  A pair the stretch cuts through is split by query reads: a pair's query volume is dealt out in
// chunks of `chunk_reads` reads, chunk c into slot c % slots, and a rank gets a contiguous range of slots (interleaving keeps the
// shares of a self pair alike, see ReadSel in stage_seed.inl).  Consequences:
//   * every rank gets the same modelled cost up to one slot of one pair;
//   * a rank's units are consecutive on the line: it needs the index of few reference volumes (usually one or two), in ascending
//     order, and a reference volume's ranks are CONSECUTIVE ranks - its team.  A team of one builds the index alone
//     (necat_index_build); a bigger team builds it in hash-range slices and all-gathers them (necat_index_build_sharded over a
//     communicator of just those ranks);
//   * records of a unit depend on nothing but the unit (a read is processed exactly as in a whole-pair run), so the union over
//     the ranks is the record set of the V jobs, and the part files of reference volume v concatenate to its pm_result_v.
// Cost model: pair (v, i) ~ bases(i) * bases(v) (hits per sampled k-mer grow with the reference volume's share of the coverage;
// both -j 0 seeding and -j 1 extension follow the candidate count), halved for i == v.  It only has to be proportional.
#pragma once
#include <stdint.h>
#include <vector>

namespace necat_host {

struct PairUnit { int32_t ref_vol, query_vol, slot_lo, slot_hi; };      // query chunks c with slot_lo <= c % slots < slot_hi

struct PairSchedule {
    int slots = 0;
    std::vector<PairUnit> units;              // all ranks' units, rank by rank
    std::vector<uint64_t> rank_off;           // [nranks + 1]: rank g owns units[rank_off[g] .. rank_off[g + 1])
    std::vector<int32_t> team_lo, team_hi;    // [V]: ranks team_lo[v] .. team_hi[v] (inclusive) work on reference volume v (lo > hi: nobody)
};

// query reads per chunk of a pair's split: 64 (the sharded calls' default), fewer for small query volumes so that every slot still
// gets several chunks; a function of the volume alone - every rank that works on a pair derives the same value
inline int pair_chunk_reads(uint64_t query_reads, int slots)
{
    const uint64_t c = query_reads / ((uint64_t)slots * 8);
    return c < 1 ? 1 : (c > 64 ? 64 : (int)c);
}

// skip (optional, [V]): reference volumes whose job is already done (oc2pm's pm<i>.finished): their pairs cost nothing and get no unit
inline PairSchedule pair_schedule(const uint64_t* vol_bases, int V, int G, int slots, const uint8_t* skip = nullptr)
{
    PairSchedule S;
    S.slots = slots;
    S.rank_off.assign((size_t)G + 1, 0);
    S.team_lo.assign((size_t)V, 0); S.team_hi.assign((size_t)V, -1);
    // costs in units that keep 128-bit products out of the way: bases / 1024
    auto kb = [&](int v) { return (long double)(vol_bases[v] / 1024 + 1); };
    long double total = 0;
    auto cost = [&](int v, int i) { return (skip && skip[v]) ? 0.0L : kb(v) * kb(i) * (i == v ? 0.5L : 1.0L); };
    for (int v = 0; v < V; ++v) for (int i = v; i < V; ++i) total += cost(v, i);
    // the slot boundary of a point x inside a pair that covers [a, b) of the line: one rounding rule for both neighbours of a cut
    auto slot_of = [&](long double x, long double a, long double b) -> int {
        if (x <= a) return 0;
        if (x >= b) return slots;
        int k = (int)((x - a) / (b - a) * (long double)slots + 0.5L);
        return k < 0 ? 0 : (k > slots ? slots : k);
    };
    for (int g = 0; g < G; ++g) {
        const long double L = total * (long double)g / (long double)G, R = g + 1 == G ? total * 2 : total * (long double)(g + 1) / (long double)G;
        long double a = 0;
        for (int v = 0; v < V; ++v) for (int i = v; i < V; ++i) {
            const long double b = a + cost(v, i);
            if (b > a && b > L && a < R) {
                const int lo = g == 0 ? 0 : slot_of(L, a, b), hi = slot_of(R, a, b);
                if (hi > lo) {
                    S.units.push_back(PairUnit{v, i, lo, hi});
                    if (S.team_hi[v] < S.team_lo[v]) S.team_lo[v] = g;
                    S.team_hi[v] = g;
                }
            }
            a = b;
        }
        S.rank_off[(size_t)g + 1] = S.units.size();
    }
    return S;
}

}  // namespace necat_host
