"""Multi-GPU work partition of the overlap stage (coarse granularity).

The unit necat.pl distributes is one reference volume (`oc2pmov ... wrk-dir i pm_result_i`,
necat.pl:190-202): volume i is mapped against every volume j >= i, so volume i costs (V - i) volume
pairs.  Ranks take reference volumes so that the pair counts balance; no data-path collective is
needed (each rank builds the index of its own reference volumes and writes its own pm_result_i).
`reduce_step_stats` is the only collective: the bench/driver bookkeeping (max time, summed counts).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def assign_reference_volumes(num_volumes: int, world: int) -> List[List[int]]:
    """Greedy longest-processing-time assignment of reference volumes (cost V - i) to ranks."""
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for v in range(num_volumes):                 # costs are already in descending order
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(v)
        loads[r] += num_volumes - v
    return out


def volume_pairs(vids: Sequence[int], num_volumes: int) -> List[Tuple[int, int]]:
    """(reference volume, query volume) pairs a rank owns (pm_worker.c:372: i = vid .. V-1)."""
    return [(v, j) for v in vids for j in range(v, num_volumes)]


def consensus_partitions(num_partitions: int, rank: int, world: int) -> List[int]:
    """Candidate partitions of the consensus stage a rank owns: the reference's own multi-node rule
    (`oc2cns ... -mn node_id num_nodes`, consensus/main.c:68-72: i = node_id; i < n; i += num_nodes).  Partitions are
    independent (every template lives in exactly one), so necat_cns_extension_batch needs no collective either."""
    return list(range(rank, num_partitions, world))


def split_templates(tmpl_off: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """One partition on several GPUs: contiguous template ranges [lo, hi) with balanced CANDIDATE counts
    (tmpl_off = the prefix offsets necat_cns_load_partition returns).  Templates are independent units of the loop."""
    n = len(tmpl_off) - 1
    total = tmpl_off[n] - tmpl_off[0]
    cuts = [0]
    for r in range(1, world):
        want = tmpl_off[0] + total * r // world
        lo, hi = cuts[-1], n
        while lo < hi:                              # first template whose offset reaches the target
            mid = (lo + hi) // 2
            if tmpl_off[mid] < want:
                lo = mid + 1
            else:
                hi = mid
        if lo > cuts[-1] and want - tmpl_off[lo - 1] < tmpl_off[lo] - want:
            lo -= 1                                 # the boundary nearest to the target
        cuts.append(lo)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def reduce_step_stats(dist, elapsed: float, overlaps: float, gbp: float, device=None):
    """(max elapsed, total overlaps, total Gbp) over all ranks; identity when dist is None."""
    if dist is None:
        return elapsed, overlaps, gbp
    import torch
    t = torch.tensor([elapsed, overlaps, gbp], dtype=torch.float64, device=device)
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(tmax[0]), float(t[1]), float(t[2])
