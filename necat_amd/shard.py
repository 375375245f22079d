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


# ---- fine granularity: ONE volume on several GPUs (necat_index_build_sharded / necat_*_sharded, include/necat_hip.h).
# The arithmetic of the C library restated on numpy arrays: what the multi-process CPU tests check, and what a launcher
# needs to reason about balance.

OFFSET_BITS = 34          # kmer_stats[h] = cnt << 34 | start   (lookup_table.h:6-21)
OFFSET_MASK = (1 << OFFSET_BITS) - 1


def index_buckets(k: int) -> Tuple[int, int]:
    """(number of hash-prefix buckets NB, log2 of the table entries per bucket) of the partitioned build: buckets of
    2^18 table entries, at most 4096 of them; (0, 0) for the small tables every rank builds whole (k < 11)"""
    pb = min(12, 2 * k - 18)
    if pb < 4:
        return 0, 0
    return 1 << pb, 2 * k - pb


def hash_range(k: int, rank: int, world: int) -> Tuple[int, int]:
    """table entries [lo, hi) rank `rank` builds: buckets [rank NB / world, (rank + 1) NB / world)"""
    nb, shift = index_buckets(k)
    if nb < world:
        return (0, 4 ** k)
    return ((rank * nb // world) << shift, ((rank + 1) * nb // world) << shift)


def index_slice(stats, offs, lo: int, hi: int):
    """What the rank that owns table entries [lo, hi) produces before the exchange: its kmer_stats slice with starts
    counted from ITS first offset, and its run of the offset list (the list is grouped by ascending hash, so a hash
    range is a contiguous run).  numpy uint64 arrays in the reference layout."""
    import numpy as np
    sl = stats[lo:hi].copy()
    cnt = sl >> np.uint64(OFFSET_BITS)
    n_before = int((stats[:lo] >> np.uint64(OFFSET_BITS)).sum())
    n_mine = int(cnt.sum())
    nz = cnt > 0
    sl[nz] -= np.uint64(n_before)
    return sl, offs[n_before:n_before + n_mine].copy()


def rebase_slice(stats_slice, base: int):
    """starts of a received kmer_stats slice moved by the number of offset entries of the ranks before its owner
    (the exclusive scan of the slice sizes); absent k-mers stay 0"""
    import numpy as np
    out = stats_slice.copy()
    nz = (out >> np.uint64(OFFSET_BITS)) > 0
    out[nz] += np.uint64(base)
    return out


def sparse_table(stats, compact_base: int = 0):
    """The sparse layout the slice build writes (IndexView, dev_common.h) of a dense kmer_stats array whose length is a multiple
    of 64: (bits[n / 64], base[n / 64], compact) - bit j of bits[w] says entry 64 w + j is non-zero, base[w] = position in
    `compact` of the word's first non-zero entry (+ compact_base: the non-zero entries of the ranks before the owner), compact =
    the non-zero entries in hash order."""
    import numpy as np
    nz = stats != 0
    w = nz.reshape(-1, 64)
    bits = (w.astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(axis=1).astype(np.uint64)
    per = w.sum(axis=1).astype(np.uint64)
    base = np.concatenate([[0], np.cumsum(per)[:-1]]).astype(np.uint64) + np.uint64(compact_base)
    return bits, base, stats[nz].copy()


def sparse_lookup(bits, base, compact, h):
    """kmer_stats[h] through the sparse layout (IndexView::lookup), vectorised over h"""
    import numpy as np
    h = np.asarray(h, dtype=np.uint64)
    w = (h >> np.uint64(6)).astype(np.int64)
    bit = np.uint64(1) << (h & np.uint64(63))
    present = (bits[w] & bit) != 0
    below = bits[w] & (bit - np.uint64(1))
    pop = np.array([bin(int(x)).count("1") for x in below], dtype=np.uint64)
    out = np.zeros(h.shape, dtype=np.uint64)
    idx = (base[w] + pop)[present].astype(np.int64)
    out[present] = compact[idx]
    return out


def read_chunks(nreads: int, chunk_reads: int, rank: int, world: int):
    """query reads of rank `rank`: chunk c (reads [c chunk, (c + 1) chunk)) belongs to rank c % world"""
    import numpy as np
    r = np.arange(nreads)
    return r[(r // chunk_reads) % world == rank]


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
