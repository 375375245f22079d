"""oc2mkdb drop-in (necat_amd/csrc/oc2mkdb_main.cpp): volume files, volume_names.txt and reads_info.txt
byte for byte as the reference writes them.  The only bytes that may differ are the 4 padding bytes at the end of each
32-byte SequenceInfo record: the reference dumps them uninitialised (common/packed_db.c:229-237 fills the fields of a
stack struct one by one), so they are masked on both sides."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from necat_amd import build, synth
from oracle import oracle_api as ora
from tests import util

MAGIC = len(synth.PAC_MAGIC)


@pytest.fixture(scope="module")
def oc2mkdb(built):
    build.build_cli()
    return build.OC2MKDB


def _masked(path):
    b = bytearray(open(path, "rb").read())
    ns = int(np.frombuffer(bytes(b[MAGIC:MAGIC + 8]), dtype="<u8")[0])
    for i in range(ns):
        o = MAGIC + 16 + 32 * i + 28
        b[o:o + 4] = b"\0\0\0\0"
    return bytes(b)


def _same_dirs(a, b):
    assert sorted(os.listdir(a)) == sorted(os.listdir(b))
    for fn in sorted(os.listdir(a)):
        if fn.startswith("vol") and fn != "volume_names.txt":
            assert _masked(os.path.join(a, fn)) == _masked(os.path.join(b, fn)), fn
        else:
            x = open(os.path.join(a, fn), "rb").read().replace(os.path.abspath(a).encode(), b"@")
            y = open(os.path.join(b, fn), "rb").read().replace(os.path.abspath(b).encode(), b"@")
            assert x == y, fn


def _tricky_inputs(d):
    rng = np.random.default_rng(5)

    def rnd(n, alphabet="ACGT"):
        return "".join(rng.choice(list(alphabet), n))
    with open(os.path.join(d, "a.fasta"), "w", newline="") as f:
        f.write(">r1 some comment here\n" + rnd(130) + "\n" + rnd(70) + "\n\n" + rnd(5) + "\n")       # multi-line, empty line
        f.write(">r2\tTABBED comment\n" + rnd(333, "ACGTacgtNn-RY") + "\n")                        # lower case, N, '-', IUPAC
        f.write(">r3\r\n" + rnd(50) + "\r\n" + rnd(21) + "\r\n")                                    # CRLF
        f.write(">empty\n>r5_after_empty x\n" + rnd(1000) + "\n")                                   # empty sequence
        f.write(">\n" + rnd(17) + "\n")                                                             # empty name
        f.write(">r7 no newline at the end\n" + rnd(64))
    with gzip.open(os.path.join(d, "b.fastq.gz"), "wt") as f:                                       # gzip FASTQ
        for i in range(40):
            n = int(rng.integers(1, 3000))
            q = "@" + "I" * (n - 1) if i % 3 == 0 else "+" * n                                      # quality lines that look like headers
            f.write("@q%d desc %d\n%s\n+%s\n%s\n" % (i, i, rnd(n, "ACGTN"), "q%d" % i if i % 2 else "", q))
    with open(os.path.join(d, "c.fastq"), "w") as f:                                                # short quality string
        f.write("@ok1\nACGTACGT\n+\nIIIIIIII\n@bad\nACGTACGTAA\n+\nIIII\n@never\nACGT\n+\nIIII\n")
    with open(os.path.join(d, "d.fa"), "w") as f:
        for i in range(120):
            s = rnd(int(rng.integers(500, 9000)))
            f.write(">big%d\n" % i)
            for k in range(0, len(s), 80):
                f.write(s[k:k + 80] + "\n")
    open(os.path.join(d, "list1.txt"), "w").write("\n".join(os.path.join(d, x) for x in ("a.fasta", "b.fastq.gz")) + "\n")
    open(os.path.join(d, "list2.txt"), "w").write("\n".join(os.path.join(d, x) for x in ("c.fastq", "d.fa")))   # no newline at the end
    return [os.path.join(d, "list1.txt"), os.path.join(d, "list2.txt")]


@pytest.mark.skipif(not ora.have_ref(), reason="oracle/_ref (reference build) not present")
def test_oc2mkdb_vs_reference(oc2mkdb, tmp_path):
    d = str(tmp_path)
    lists = _tricky_inputs(d)
    ref_exe = os.path.join(os.path.dirname(ora.REF_PMOV), "oc2mkdb")
    for exe, out in ((ref_exe, "ref"), (oc2mkdb, "mine")):
        r = subprocess.run([exe, os.path.join(d, out)] + lists, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
    _same_dirs(os.path.join(d, "ref"), os.path.join(d, "mine"))
    nv, nr = [int(x) for x in open(os.path.join(d, "mine", "reads_info.txt")).read().split()]
    assert (nv, nr) == (1, 7 + 40 + 2 + 120)


@pytest.mark.parametrize("ds", ["vols_a", "vols_b"])
def test_oc2mkdb_reproduces_golden_volumes(oc2mkdb, tmp_path, ds):
    """FASTA written from the golden volumes' own reads -> the golden volume files (these were checked against the
    reference's oc2mkdb when they were generated, tests/golden/make_golden.py); several volumes via the size override"""
    src = util.install_golden_volumes(ds, tmp_path)
    from necat_amd import capi
    nv, nr, vols = capi.load_volumes_info(src)
    fa = os.path.join(str(tmp_path), "reads.fasta")
    first_sizes = []
    with open(fa, "w") as f:
        for path, start, cnt in vols:
            pac, off, sz, names = synth.read_volume(path)
            codes = synth.unpack_2bit(pac, int(sz.sum()))
            first_sizes.append(int(sz.sum()))
            for i in range(len(names)):
                s = "ACGT".encode()
                seq = bytes(s[c] for c in codes[int(off[i]):int(off[i] + sz[i])])
                f.write(">%s\n%s\n" % (names[i], seq.decode()))
    lst = os.path.join(str(tmp_path), "list.txt")
    open(lst, "w").write(fa + "\n")
    out = os.path.join(str(tmp_path), "out")
    env = dict(os.environ)
    if nv > 1:
        # the golden multi-volume set was cut at a small volume size: a volume closes once it holds >= that many bases
        man_size = min(first_sizes[:-1])
        env["NECAT_MKDB_VOLSIZE"] = str(man_size - min(200, man_size // 2))
    r = subprocess.run([oc2mkdb, out, lst], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    assert r.returncode == 0, r.stdout
    nv2, nr2, vols2 = capi.load_volumes_info(out)
    assert nr2 == nr
    if nv2 == nv:
        for (p1, s1, c1), (p2, s2, c2) in zip(vols, vols2):
            assert (s1, c1) == (s2, c2)
            assert _masked(p1) == _masked(p2)
    else:
        pytest.skip("volume boundaries of the golden set are not reproducible from one size threshold")


@pytest.mark.gpu
def test_oc2mkdb_device_packing_equals_host_packing(oc2mkdb, tmp_path):
    """NECAT_MKDB_GPU=1: the 2-bit packing of a volume runs on the GPU (necat_volume_pack).  On the tricky inputs - lower case,
    N, '-', IUPAC codes whose nst_nt4 codes 4 / 5 spill into the neighbouring 2-bit slot, sequences that start in the middle of
    a pac byte, several volumes - every output file equals the host-packed one (which test_oc2mkdb_vs_reference pins to the
    reference's own oc2mkdb)."""
    d = str(tmp_path)
    lists = _tricky_inputs(d)
    for out, env in (("host", {}), ("gpu", {"NECAT_MKDB_GPU": "1"})):
        r = subprocess.run([oc2mkdb, os.path.join(d, out)] + lists, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           env=dict(os.environ, NECAT_MKDB_VOLSIZE="150000", **env))
        assert r.returncode == 0, r.stdout
    _same_dirs(os.path.join(d, "host"), os.path.join(d, "gpu"))
    assert int(open(os.path.join(d, "gpu", "reads_info.txt")).read().split()[0]) >= 3
    # and directly against the REFERENCE's own oc2mkdb (oracle/_ref travels to the GPU box): one volume (its 2 Gbp cut is a constant)
    ref_exe = os.path.join(os.path.dirname(ora.REF_PMOV), "oc2mkdb")
    if os.path.exists(ref_exe):
        r = subprocess.run([ref_exe, os.path.join(d, "ref")] + lists, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        r = subprocess.run([oc2mkdb, os.path.join(d, "gpu1")] + lists, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, NECAT_MKDB_GPU="1"))
        assert r.returncode == 0, r.stdout
        _same_dirs(os.path.join(d, "ref"), os.path.join(d, "gpu1"))


@pytest.mark.gpu
def test_volume_pack_gives_the_uploaded_volume(ctx, tmp_path):
    """necat_volume_pack(text) = necat_volume_upload(host-packed pac): the same index comes out of both volumes"""
    import ctypes as C
    from necat_amd import capi
    rng = np.random.default_rng(9)
    sizes = rng.integers(300, 4000, 60).astype(np.uint64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    codes = rng.integers(0, 4, int(sizes.sum()), dtype=np.uint8)
    text = bytes(b"ACGT"[c] for c in codes)
    pac = np.zeros((len(text) + 3) // 4 + 8, dtype=np.uint8)
    vol = C.c_void_p()
    rc = ctx.lib.necat_volume_pack(ctx.h, text, len(text), offs.ctypes.data_as(C.c_void_p), sizes.ctypes.data_as(C.c_void_p), len(sizes),
                                   pac.ctypes.data_as(C.c_void_p), C.byref(vol))
    assert rc == 0, ctx.lib.necat_last_error(ctx.h).decode()
    want = synth.pack_2bit(codes)
    assert pac[:want.shape[0]].tobytes() == want.tobytes()
    v2 = ctx.upload_volume(want, len(text), offs, sizes)
    ix2 = ctx.build_index(v2, 11, 100)
    a2 = ix2.download()
    ix2.free(); v2.free()
    v1 = capi.Volume(ctx, vol, len(text), offs.astype(np.int64), sizes.astype(np.int64))
    ix1 = ctx.build_index(v1, 11, 100)
    a1 = ix1.download()
    ix1.free(); v1.free()
    assert all(np.array_equal(x, y) for x, y in zip(a1, a2))
