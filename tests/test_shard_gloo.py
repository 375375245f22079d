"""CPU tests of the N > 1 path: volume partition + the world_size-2 gloo reduction bench.py uses."""
import os
import socket

import pytest

from necat_amd import shard


def test_assignment_covers_every_volume_once_and_balances():
    for V in (1, 3, 6, 45):
        for W in (1, 2, 4, 8):
            a = shard.assign_reference_volumes(V, W)
            flat = sorted(v for r in a for v in r)
            assert flat == list(range(V))
            loads = [sum(V - v for v in r) for r in a]
            if V >= W:
                assert max(loads) - min(loads) <= V          # within one volume's cost
    pairs = shard.volume_pairs([0, 2], 3)
    assert pairs == [(0, 0), (0, 1), (0, 2), (2, 2)]


def test_consensus_units_partition_exactly():
    for n in (0, 1, 7, 100):
        for w in (1, 2, 8):
            got = sorted(p for r in range(w) for p in shard.consensus_partitions(n, r, w))
            assert got == list(range(n))
    off = [0, 5, 5, 30, 31, 60, 100, 100, 130]
    for w in (1, 2, 3, 8, 20):
        rs = shard.split_templates(off, w)
        assert rs[0][0] == 0 and rs[-1][1] == len(off) - 1
        assert all(a[1] == b[0] for a, b in zip(rs, rs[1:])) and all(lo <= hi for lo, hi in rs)
    lo, hi = shard.split_templates(off, 2)[0]
    assert abs((off[hi] - off[lo]) - 65) <= 30          # about half of the 130 candidates


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vols = shard.assign_reference_volumes(5, world)[rank]
    pairs = shard.volume_pairs(vols, 5)
    out = shard.reduce_step_stats(dist, 1.0 + rank, float(len(pairs)), 0.5 * (rank + 1))
    q.put((rank, out, len(pairs)))
    dist.destroy_process_group()


def test_world_size_2_reduction():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    total_pairs = sum(r[2] for r in res)
    assert total_pairs == 15                     # V (V + 1) / 2 for V = 5
    for rank, (tmax, n, gbp), _ in res:
        assert tmax == 2.0 and n == 15.0 and gbp == 1.5
