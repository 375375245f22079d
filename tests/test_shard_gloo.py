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


# ---- fine granularity: one volume on several ranks (the arithmetic of necat_index_build_sharded on real arrays) ----

def _index_worker(rank, world, port, vol_path, k, q):
    """Each rank cuts ITS hash-range slice out of the single-rank index (= what its local build produces: starts counted
    from its own first offset), the ranks all-gather slice sizes, slices and offset runs over gloo (through the very
    callback glue bench.py hands to the C library), rebase the starts by the exclusive scan of the sizes, and must end
    up with the single-rank index again."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from necat_amd import dist as ndist
    from oracle import oracle_api as ora
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    allgather = ndist.torch_allgather(dist)
    stats, offs = ora.build_index(vol_path, k, 500)
    lo, hi = shard.hash_range(k, rank, world)
    my_stats, my_offs = shard.index_slice(stats, offs, lo, hi)
    # 1. slice sizes (the host all-gather of 8 bytes per rank the library does mid-build)
    sizes = [int(np.frombuffer(b, dtype=np.uint64)[0]) for b in allgather(np.uint64(my_offs.shape[0]).tobytes())]
    base = [sum(sizes[:g]) for g in range(world)]
    # 2. the owner writes final starts (the library adds its base inside k_slice_emit), then the slices travel
    mine_final = shard.rebase_slice(my_stats, base[rank])
    parts = allgather(mine_final.tobytes()) if len(set(hi2 - lo2 for lo2, hi2 in (shard.hash_range(k, g, world) for g in range(world)))) == 1 else None
    if parts is None:      # unequal slices (world does not divide the bucket count): gather-v through object lists
        objs = [None] * world
        dist.all_gather_object(objs, mine_final.tobytes())
        parts = objs
    full_stats = np.concatenate([np.frombuffer(b, dtype=np.uint64) for b in parts])
    objs = [None] * world
    dist.all_gather_object(objs, my_offs.tobytes())
    full_offs = np.concatenate([np.frombuffer(b, dtype=np.uint64) for b in objs])
    ok = bool(np.array_equal(full_stats, stats) and np.array_equal(full_offs, offs))
    # 3. the same exchange in the layout the library really moves (necat_index_build_sharded): the owner's IdxWords with FINAL
    # compact bases (its first non-zero entry sits behind those of the ranks before it) + its run of non-zero entries
    n_nz = [int(np.frombuffer(b, dtype=np.uint64)[0]) for b in allgather(np.uint64(int((my_stats != 0).sum())).tobytes())]
    bits, cbase, comp = shard.sparse_table(mine_final, sum(n_nz[:rank]))
    objs = [None] * world
    dist.all_gather_object(objs, (bits.tobytes(), cbase.tobytes(), comp.tobytes()))
    g_bits = np.concatenate([np.frombuffer(o[0], dtype=np.uint64) for o in objs])
    g_base = np.concatenate([np.frombuffer(o[1], dtype=np.uint64) for o in objs])
    g_comp = np.concatenate([np.frombuffer(o[2], dtype=np.uint64) for o in objs])
    w_bits, w_base, w_comp = shard.sparse_table(stats)
    ok = ok and bool(np.array_equal(g_bits, w_bits) and np.array_equal(g_base, w_base) and np.array_equal(g_comp, w_comp))
    probe = np.random.default_rng(rank).integers(0, stats.shape[0], 4000).astype(np.uint64)
    ok = ok and bool(np.array_equal(shard.sparse_lookup(g_bits, g_base, g_comp, probe), stats[probe.astype(np.int64)]))
    # the slices tile the table and the reads
    covered = sum(h - l for l, h in (shard.hash_range(k, g, world) for g in range(world)))
    reads = np.concatenate([shard.read_chunks(1000, 64, g, world) for g in range(world)])
    q.put((rank, ok, sizes, covered == 4 ** k, sorted(reads.tolist()) == list(range(1000)), int(stats.shape[0])))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,k", [(2, 12), (3, 11)])
def test_sharded_index_arithmetic_over_gloo(tmp_path, world, k):
    import torch.multiprocessing as mp
    from tests import util
    d, rs, nv = util.make_dataset(tmp_path, genome=60_000, coverage=10.0, seed=9)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_index_worker, args=(r, world, port, os.path.join(d, "vol0"), k, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, sizes, tiles, reads_ok, T in res:
        assert ok, "rank %d: gathered index differs from the single-rank one" % rank
        assert tiles and reads_ok and T == 4 ** k
        assert sum(sizes) > 10_000 and min(sizes) > 0


def _file_ag_worker(d, r, n, tag, q):
    from necat_amd import dist as ndist
    ag = ndist.file_allgather(d, r, n, timeout_s=30)
    q.put((r, [ag(("%s-%d-%d" % (tag, r, k)).encode()) for k in range(5)]))


def test_file_allgather_rerun_in_the_same_directory(tmp_path):
    """the launcher-less all-gather (files in a shared directory) run twice in ONE directory: the second run must not read the
    first run's files - same names, same lengths, other contents (stale IPC handles / ncclUniqueIds in real use)"""
    import multiprocessing as mp
    ctxm = mp.get_context("spawn")
    for tag in ("A", "B"):
        q = ctxm.Queue()
        ps = [ctxm.Process(target=_file_ag_worker, args=(str(tmp_path), r, 3, tag, q)) for r in range(3)]
        [p.start() for p in ps]
        res = [q.get(timeout=120) for _ in ps]
        [p.join() for p in ps]
        for r, out in res:
            for k, parts in enumerate(out):
                assert parts == [("%s-%d-%d" % (tag, x, k)).encode() for x in range(3)]


def test_rccl_branch_of_comm_h_with_a_fake_rccl(tmp_path):
    """tests/host_core/check_comm.cpp: the SOURCE of necat_amd/csrc/comm.h (allgatherv_inplace, gatherv, agree) compiled with g++, the ranks
    as threads, ncclSend / ncclRecv / group calls replaced by an in-process mailbox that logs every call - peer order, byte counts, empty parts,
    root != 0 at world 1, 2, 3, 8, and a rank whose ncclSend fails (the group is closed, every rank gets a verdict, nobody blocks).  The RCCL
    transport itself has only ever run with ranks on ONE device here (1-GPU boxes): this is what pins its call pattern."""
    import subprocess
    exe = os.path.join(str(tmp_path), "check_comm")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe,
                        os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_core", "check_comm.cpp"), "-lpthread", "-ldl"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "ok" in r.stdout
