"""oc2pcan drop-in (necat_amd/csrc/oc2pcan_main.cpp): the same partition files as the reference's
candidate partitioner - same file set, same `.partitions` count, same records per partition (their order inside a
file is unspecified in the reference: worker threads append chunks)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from necat_amd import build, capi
from oracle import oracle_api as ora
from tests import util


@pytest.fixture(scope="module")
def oc2pcan(built):
    build.build_cli()
    return build.OC2PCAN


def _records(path):
    b = open(path, "rb").read()
    assert len(b) % 28 == 0
    return sorted(b[i:i + 28] for i in range(0, len(b), 28))


def _candidates(tmp_path, ds):
    """all-vs-all candidates of a golden data set by the oracle's oc2pmov -j 0 -u 1, every volume"""
    wrk = util.install_golden_volumes(ds, tmp_path)
    nv, nr, _ = capi.load_volumes_info(wrk)
    o = ora.options(**dict(util.FAST, kmer_size=13, job=0, binary_output=1, num_threads=2))
    rec = b""
    for v in range(nv):
        out = os.path.join(str(tmp_path), "pm_%d" % v)
        ora.pm_main(o, v, wrk, out)
        rec += open(out, "rb").read()
    return wrk, nr, rec


@pytest.mark.skipif(not ora.have_ref_cns(), reason="oracle/_ref (reference build) not present")
@pytest.mark.parametrize("ds,args", [("vols_a", []), ("vols_b", ["-p", "7"]), ("vols_b", ["-p", "5", "-f", "3"]), ("vols_a", ["-p", "1000000", "-t", "4"])])
def test_oc2pcan_vs_reference(oc2pcan, tmp_path, ds, args):
    wrk, nr, rec = _candidates(tmp_path, ds)
    assert len(rec) > 28 * 50
    outs = {}
    for tag, exe in (("ref", ora.REF_PCAN), ("mine", oc2pcan)):
        d = os.path.join(str(tmp_path), tag)
        os.makedirs(d)
        can = os.path.join(d, "cands")
        open(can, "wb").write(rec)
        r = subprocess.run([exe] + args + [wrk, can], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        outs[tag] = d
    assert sorted(os.listdir(outs["ref"])) == sorted(os.listdir(outs["mine"]))
    assert open(os.path.join(outs["ref"], "cands.partitions")).read() == open(os.path.join(outs["mine"], "cands.partitions")).read()
    n_parts = int(open(os.path.join(outs["mine"], "cands.partitions")).read())
    total = 0
    for p in range(n_parts):
        a, b = _records(os.path.join(outs["ref"], "cands.p%d" % p)), _records(os.path.join(outs["mine"], "cands.p%d" % p))
        assert a == b, p
        total += len(b)
    assert total == 2 * (len(rec) // 28)            # every read id is inside some batch: each record lands twice


def test_oc2pcan_single_partition_is_role_swap(oc2pcan, tmp_path):
    """no reference build needed: with one batch the partition is the input plus its role-swapped twin
    (capi.pcan_single_partition, itself pinned to the reference in tests/test_oracle_golden.py)"""
    wrk = util.install_golden_volumes("vols_a", tmp_path)
    rec = open(os.path.join(util.GOLDEN, "a_fast_can_bin.bin"), "rb").read()
    can = os.path.join(str(tmp_path), "cands")
    open(can, "wb").write(rec)
    r = subprocess.run([oc2pcan, wrk, can], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert open(can + ".partitions").read() == "1\n"
    want = capi.pcan_single_partition(rec)
    assert _records(can + ".p0") == sorted(want[i:i + 28] for i in range(0, len(want), 28))
    # a missing work directory is an error, not an empty result
    r = subprocess.run([oc2pcan, os.path.join(str(tmp_path), "nowhere"), can], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0


@pytest.mark.gpu
@pytest.mark.parametrize("ds,batch", [("vols_b", 100), ("vols_b", 37), ("vols_a", 100000)])
def test_oc2pm_writes_the_partitions_oc2pcan_would(oc2pcan, tmp_path, ds, batch):
    """NECAT_PM_PARTITIONS=<batch size>: `oc2pm -j 0 -u 1` partitions the candidates of every job on the device
    (necat_pcan_partition) and leaves <output>.p<i> + <output>.partitions next to <output>.  They must hold what oc2pcan makes
    from <output> (same file set, same records per partition; the order inside a file is free in the reference too)."""
    pmov, pm = build.build_cli()
    wrk = util.install_golden_volumes(ds, tmp_path)
    o = ora.options(**dict(util.FAST, kmer_size=13, job=0, binary_output=1, num_threads=2))
    out = os.path.join(str(tmp_path), "cands.bin")
    r = subprocess.run([pm] + ora.opt_argv(o) + [wrk, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, NECAT_PM_PARTITIONS=str(batch)))
    assert r.returncode == 0, r.stderr
    ref = os.path.join(str(tmp_path), "ref")
    os.makedirs(ref)
    shutil.copy(out, os.path.join(ref, "cands.bin"))
    # the partitioner to compare with: the REFERENCE's own oc2pcan when oracle/_ref is there (it travels to the GPU box), else this repo's
    exe = ora.REF_PCAN if os.path.exists(ora.REF_PCAN) else oc2pcan
    r = subprocess.run([exe, "-p", str(batch), "-f", "7", wrk, os.path.join(ref, "cands.bin")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    n = int(open(os.path.join(ref, "cands.bin.partitions")).read())
    assert int(open(out + ".partitions").read()) == n and n >= 1
    total = 0
    for p in range(n):
        want = _records(os.path.join(ref, "cands.bin.p%d" % p))
        got = _records(out + ".p%d" % p) if os.path.exists(out + ".p%d" % p) else []
        assert got == want, "partition %d" % p
        total += len(got)
    assert total > 100
    assert not [f for f in os.listdir(wrk) if f.startswith("pm_result_")]


@pytest.mark.gpu
@pytest.mark.parametrize("batch,nreads", [(1, 5000), (7, 5000), (100000, 5000)])
def test_pcan_partition_many_partitions(ctx, batch, nreads):
    """necat_pcan_partition against a numpy statement of pcan.c:47-75 on random candidates: more partitions than the kernel counts
    in LDS (batch 1: 5000 partitions, global counters), a few hundred, and one; ids outside [0, num_reads) go nowhere."""
    import ctypes as C
    rng = np.random.default_rng(3)
    n = 40000
    c = np.zeros(n, dtype=capi.CANDIDATE_DTYPE)
    c["qid"] = rng.integers(-3, nreads + 5, n); c["sid"] = rng.integers(-3, nreads + 5, n)
    c["qdir"] = rng.integers(0, 2, n); c["sdir"] = rng.integers(0, 2, n); c["score"] = rng.integers(1, 2000000, n)
    for f in ("qbeg", "qend", "sbeg", "send", "qoff", "soff"):
        c[f] = rng.integers(0, 50000, n)
    c["qoff"][::3] = c["qbeg"][::3]
    rec, off, npart = C.c_void_p(), C.c_void_p(), C.c_int()
    rc = ctx.lib.necat_pcan_partition(ctx.h, c.ctypes.data_as(C.c_void_p), n, batch, nreads, C.byref(rec), C.byref(off), C.byref(npart))
    assert rc == 0, ctx.lib.necat_last_error(ctx.h).decode()
    want_parts = (nreads + batch - 1) // batch
    assert npart.value == want_parts
    poff = np.ctypeslib.as_array(C.cast(off, C.POINTER(C.c_uint64)), shape=(want_parts + 1,)).copy()
    total = int(poff[-1])
    got = np.ctypeslib.as_array(C.cast(rec, C.POINTER(C.c_uint32)), shape=(max(total, 1) * 7,))[: total * 7].reshape(-1, 7).copy()
    ctx.lib.necat_free(rec); ctx.lib.necat_free(off)
    a = capi.pack_candidates(c).astype(np.uint32)
    b = np.frombuffer(capi.pcan_single_partition(a.tobytes()), dtype="<u4").reshape(-1, 7)[n:]        # the role-swapped twins
    allrec = np.concatenate([a, b])
    tid = allrec[:, 1].astype(np.int32).astype(np.int64)
    part = np.where((tid >= 0) & (tid // batch < want_parts), tid // batch, -1)
    assert total == int((part >= 0).sum())
    for p in np.unique(np.concatenate([rng.integers(0, want_parts, 40), [0, want_parts - 1]])):
        mine = sorted(map(bytes, got[int(poff[p]):int(poff[p + 1])]))
        want = sorted(map(bytes, allrec[part == p]))
        assert mine == want, "partition %d" % p
    counts = np.bincount(part[part >= 0], minlength=want_parts)
    assert np.array_equal(np.diff(poff).astype(np.int64), counts)
