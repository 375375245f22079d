"""One reference volume on several ranks (fine granularity): hash-range sharded index build + all-gather,
query chunks dealt out to the ranks, records gathered on rank 0.  The ranks here share device 0 (a 1-GPU box), so the
device memory moves by the HIP IPC transport; everything else - slice arithmetic, rebased starts, read selection, gather-v -
is the code the RCCL transport runs too.  Checked against the single-rank result of the same library (which the other GPU
tests pin to the oracle) and, for the index, directly against the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util
from oracle import oracle_api as ora

pytestmark = pytest.mark.gpu
WORKER = os.path.join(util.ROOT, "tests", "tools", "shard_worker.py")


def _run_ranks(nranks, vdir, tmp, k, z, chunk=16, shard="1"):
    # shard: NECAT_INDEX_SHARD - "1" hash-range slices + all-gather, "0" every rank builds the whole table (what necat_index_plan picks for a volume
    # this small, and for E. coli / yeast at N <= 4), "" the plan's own choice
    xdir = os.path.join(str(tmp), "xchg_%d_%d_%s" % (nranks, k, shard))
    os.makedirs(xdir)
    prefix = os.path.join(str(tmp), "r%d_k%d_%s" % (nranks, k, shard))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SHARD_CHUNK=str(chunk))
    env.pop("NECAT_INDEX_SHARD", None)
    if shard != "":
        env["NECAT_INDEX_SHARD"] = shard
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(nranks), "0", vdir, xdir, prefix, str(k), str(z), "auto"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(nranks)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    return prefix


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d, rs, nv = util.make_dataset(tmp_path_factory.mktemp("shard"), genome=150_000, coverage=18.0, seed=5)
    return d, rs


@pytest.mark.parametrize("nranks,k,shard", [(2, 13, "1"), (3, 12, "1"), (2, 13, "0"), (3, 12, "")])
def test_sharded_run_equals_single_rank(ctx, data, tmp_path, nranks, k, shard):
    from necat_amd import capi
    d, rs = data
    kw = dict(util.FAST, kmer_size=k)
    prefix = _run_ranks(nranks, d, tmp_path, k, kw["scan_window"], shard=shard)
    # ---- the gathered index: complete and identical on every rank, equal to the oracle's
    ostats, ooffs = ora.build_index(os.path.join(d, "vol0"), k, kw["kmer_cnt_cutoff"])
    for r in range(nranks):
        assert np.array_equal(np.load(prefix + "_stats_%d.npy" % r), ostats), "kmer_stats of rank %d" % r
        assert np.array_equal(np.load(prefix + "_offs_%d.npy" % r), ooffs), "offset_list of rank %d" % r
    infos = [json.load(open(prefix + "_info_%d.json" % r)) for r in range(nranks)]
    assert all(i["transport"] == "ipc" for i in infos)                       # the ranks share device 0
    if shard == "1":
        assert all(i["index_exchange_bytes"] > 0 and i["index_sharded"] == 1 for i in infos)
    else:       # replicate mode (forced, or the plan's choice for a 2.7 Mbp volume): no rank received a byte of index
        assert all(i["index_exchange_bytes"] == 0 and i["index_sharded"] == 0 for i in infos)
    assert sum(i["reads_local"] for i in infos) == rs.nreads
    assert min(i["reads_local"] for i in infos) > 0
    # ---- records: rank 0 holds everybody's, equal to the single-rank run
    vol = ctx.load_volume(os.path.join(d, "vol0"))
    ix = ctx.build_index(vol, k, kw["kmer_cnt_cutoff"])
    c1 = ctx.find_candidates(ix, vol, vol, 0, 0, capi.default_options(**dict(kw, job=0)), True)
    m1, _ = ctx.map_pair(ix, vol, vol, 0, 0, capi.default_options(**dict(kw, job=1)), True, 1)
    ix.free(); vol.free()
    c0 = np.load(prefix + "_cands_0.npy")
    m0 = np.load(prefix + "_m4_0.npy")
    assert c0.shape[0] == c1.shape[0] == sum(i["cands_local"] for i in infos) and c1.shape[0] > 500
    assert sorted(bytes(r) for r in capi.pack_candidates(c0).astype("<u4")) == sorted(bytes(r) for r in capi.pack_candidates(c1).astype("<u4"))
    assert m0.shape[0] == m1.shape[0] == sum(i["m4_local"] for i in infos) and m1.shape[0] > 500
    assert util.m4_key_rows(m0) == util.m4_key_rows(m1)
    # the other ranks return their own records only
    for r in range(1, nranks):
        assert np.load(prefix + "_m4_%d.npy" % r).shape[0] == infos[r]["m4_local"]


def test_sharded_k15_records(ctx, data, tmp_path):
    """k = 15: the table of 2^30 entries in 4096 buckets, two ranks exchange half of it each (sparse layout: 16 bytes per 64
    entries + the non-zero entries); records equal the single-rank ones"""
    from necat_amd import capi
    d, rs = data
    kw = dict(util.FAST, kmer_size=15)
    prefix = _run_ranks(2, d, tmp_path, 15, kw["scan_window"], chunk=64)
    vol = ctx.load_volume(os.path.join(d, "vol0"))
    ix = ctx.build_index(vol, 15, kw["kmer_cnt_cutoff"])
    m1, _ = ctx.map_pair(ix, vol, vol, 0, 0, capi.default_options(**dict(kw, job=1)), True, 1)
    ix.free(); vol.free()
    m0 = np.load(prefix + "_m4_0.npy")
    assert m1.shape[0] > 300 and util.m4_key_rows(m0) == util.m4_key_rows(m1)
    info = json.load(open(prefix + "_info_0.json"))
    assert info["index_exchange_bytes"] >= (1 << 30) // 64 * 16 // 2        # half of the table's words came from the other rank


def test_rccl_transport_call_path_runs(ctx):
    """The RCCL transport cannot be exercised between ranks on a 1-GPU box (RCCL refuses two ranks on one device; the multi-rank
    tests above use HIP IPC).  Its call path - librccl opened at run time, ncclGetUniqueId / ncclCommInitRank, an
    ncclSend / ncclRecv group on the context's stream - runs here with one rank sending to itself."""
    rc = ctx.lib.necat_comm_selftest_rccl(ctx.h, 1 << 20)
    assert rc == 0, ctx.lib.necat_last_error(ctx.h).decode()


def test_rccl_exchange_between_two_devices(ctx):
    """The RCCL data path between two DEVICES (what comm.h's all-pairs exchange is made of): two ranks of one process, each sends to and receives
    from the other over the link.  Skipped - with the reason - on a box with one device; the first multi-GPU box to run the suite runs it."""
    rc = ctx.lib.necat_comm_selftest_rccl2(ctx.h, 4 << 20)
    if rc == 1:
        pytest.skip("one device on this box: " + ctx.lib.necat_last_error(ctx.h).decode())
    assert rc == 0, ctx.lib.necat_last_error(ctx.h).decode()
