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
