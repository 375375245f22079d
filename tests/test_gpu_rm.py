"""GPU parity of the read-to-reference mapper (second half of SURVEY 8f.4): the oc2rm_worker program and necat_map_reference through
the C ABI against the REFERENCE's own oc2rm_worker -t 1 (oracle/_ref/oc2rm_worker, built from /root/reference; it travels to the
GPU box) - text records with ids and with names, 96-byte binary records, the -mn volume split."""
import os
import subprocess

import numpy as np
import pytest

from necat_amd import build, capi
from oracle import oracle_api as ora
from tests import util

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not os.path.exists(ora.REF_RM), reason="needs oracle/_ref/oc2rm_worker")


def _ref(args, wrk, ref, out, mn=None):
    subprocess.run([ora.REF_RM] + args + ["-t", "1", wrk, ref, out] + (["-mn", str(mn[0]), str(mn[1])] if mn else []), check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return open(out, "rb").read()


def _mine(built, args, wrk, ref, out, mn=None):
    built.build_cli()
    r = subprocess.run([build.OC2RM] + args + ["-t", "4", wrk, ref, out] + (["-mn", str(mn[0]), str(mn[1])] if mn else []),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    return open(out, "rb").read()


def test_oc2rm_worker_golden(built, tmp_path):
    """the committed vectors: tests/golden/rm_e/ref.m4 = what the reference's oc2rm_worker wrote for tests/golden/vols_e against rm_e/ref.vol"""
    import json
    m = json.load(open(os.path.join(util.GOLDEN, "manifest_asm_rm.json")))["rm_e"]
    wrk = util.install_golden_volumes(m["volumes"], tmp_path)
    got = _mine(built, m["args"].split(), wrk, os.path.join(util.GOLDEN, "rm_e", "ref.vol"), os.path.join(str(tmp_path), "mine.m4"))
    assert got == open(os.path.join(util.GOLDEN, "rm_e", "ref.m4"), "rb").read()


@needs_ref
@pytest.mark.parametrize("seed,repeat,args", [
    (13, 0.6, "-k 13 -i 0"),
    (12, 0.4, "-k 12 -z 10 -n 8 -a 1000"),          # names as ids (the default)
    (17, 0.2, "-k 13 -b 2000 -e 0.3 -u 1"),         # binary records
])
def test_oc2rm_worker_reproduces_reference(built, tmp_path, seed, repeat, args):
    wrk, ref, nv = util.make_rm_dataset(tmp_path, seed=seed, repeat_frac=repeat)
    want = _ref(args.split(), wrk, ref, os.path.join(str(tmp_path), "ref.m4"))
    got = _mine(built, args.split(), wrk, ref, os.path.join(str(tmp_path), "mine.m4"))
    assert len(want) > 5000
    if "-u 1" in args:          # 96-byte records: the reference writes whatever its stack held into the 4 padding bytes
        a, b = np.frombuffer(got, dtype=capi.M4_DTYPE), np.frombuffer(want, dtype=capi.M4_DTYPE)
        assert a.shape == b.shape
        for f in capi.M4_DTYPE.names:
            if not f.startswith("_"):
                assert (a[f] == b[f]).all(), f
    else:
        assert got == want


@needs_ref
def test_oc2rm_worker_node_split_and_abi(ctx, built, tmp_path):
    wrk, ref, nv = util.make_rm_dataset(tmp_path, seed=21, repeat_frac=0.5)
    assert nv >= 3
    args = "-k 13 -i 0".split()
    parts = []
    for node in range(2):
        want = _ref(args, wrk, ref, os.path.join(str(tmp_path), "ref_%d.m4" % node), mn=(node, 2))
        got = _mine(built, args, wrk, ref, os.path.join(str(tmp_path), "mine_%d.m4" % node), mn=(node, 2))
        assert got == want and len(want) > 1000
        parts.append(got)
    whole = _mine(built, args, wrk, ref, os.path.join(str(tmp_path), "mine_all.m4"))
    assert sorted((parts[0] + parts[1]).splitlines()) == sorted(whole.splitlines())
    # the same through the C ABI, volume by volume
    nvol, nreads, vols = capi.load_volumes_info(wrk)
    opt = capi.default_options(kmer_size=13, scan_window=5, kmer_cnt_cutoff=500, block_size=1000, block_score_cutoff=3, num_candidates=20,
                               align_size_cutoff=400, ddfs_cutoff=0.25, error=0.5, num_output=20, job=1)
    rv = ctx.load_volume(ref)
    ix = ctx.build_index(rv, 13, 500)
    lines, rescued = [], 0
    for path, start, _ in vols:
        v = ctx.load_volume(path)
        m4, ncand, nresc = ctx.map_reference(ix, rv, v, int(start), 0, opt)
        rescued += nresc
        assert ncand >= m4.shape[0]
        for m in m4:
            lines.append(b"%d\t%d\t%.2f\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d" % (m["qid"], m["sid"], m["ident_perc"], m["vscore"], m["qdir"], m["qoff"], m["qend"],
                                                                               m["qsize"], m["sdir"], m["soff"], m["send"], m["ssize"]))
        v.free()
    ix.free()
    rv.free()
    assert rescued > 5
    assert lines == whole.splitlines()
