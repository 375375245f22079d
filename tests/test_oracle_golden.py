"""CPU tests: the oracle (oracle/necat_oracle.c) against the golden vectors generated from the
compiled reference (tests/golden/make_golden.py), and - when oracle/_ref is present - against the
reference itself on fresh seeded data."""
import hashlib
import json
import os
import shutil

import pytest

from tests import util
from oracle import oracle_api as ora

MANIFEST = json.load(open(os.path.join(util.GOLDEN, "manifest.json")))


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_oracle_reproduces_golden(name, tmp_path, built):
    m = MANIFEST[name]
    d = util.install_golden_volumes(m["dataset"], tmp_path)
    o = ora.options(**m["options"])
    out = os.path.join(str(tmp_path), "o.out")
    st = ora.pm_main(o, m["vid"], d, out)
    recs = ora.sorted_records(out, ora.record_size(o))
    gold = open(os.path.join(util.GOLDEN, m["file"]), "rb").read()
    assert st.n_records == m["records"] == len(recs)
    assert hashlib.md5(gold).hexdigest() == m["md5"]
    assert b"".join(recs) == gold


@pytest.mark.skipif(not ora.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("job,binary", [(0, 1), (1, 0)])
def test_oracle_matches_reference_on_fresh_data(job, binary, tmp_path, built):
    d, rs, nv = util.make_dataset(tmp_path, genome=120_000, coverage=15.0, seed=77, err=0.13, vol_size=900_000)
    assert nv >= 2
    kw = dict(util.SENSITIVE, kmer_size=12, job=job, binary_output=binary)
    for vid in range(nv):
        o = ora.options(**kw)
        a = os.path.join(str(tmp_path), "ref.out")
        b = os.path.join(str(tmp_path), "ora.out")
        ora.run_ref(o, vid, d, a)
        ora.pm_main(o, vid, d, b)
        ra, rb = ora.sorted_records(a, 28 if binary else 0), ora.sorted_records(b, 28 if binary else 0)
        assert len(ra) > 0 and ra == rb


@pytest.mark.skipif(not ora.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_volume_writer_matches_oc2mkdb(tmp_path, built):
    """necat_amd.synth.write_volume_dir against the reference's oc2mkdb (struct padding ignored)."""
    import subprocess
    import numpy as np
    from necat_amd import synth
    rs = synth.simulate_reads(30_000, 8.0, seed=5)
    d = os.path.join(str(tmp_path), "mine")
    synth.write_volume_dir(d, rs)
    fa = os.path.join(str(tmp_path), "r.fa")
    synth.write_fasta(fa, rs)
    lst = os.path.join(str(tmp_path), "list.txt")
    open(lst, "w").write(fa + "\n")
    refd = os.path.join(str(tmp_path), "ref")
    subprocess.run([ora.REF_MKDB, refd, lst], check=True, stdout=subprocess.DEVNULL)
    a = synth.read_volume(os.path.join(d, "vol0"))
    b = synth.read_volume(os.path.join(refd, "vol0"))
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)
    assert a[3] == b[3]
    assert open(os.path.join(d, "reads_info.txt")).read() == open(os.path.join(refd, "reads_info.txt")).read()


def test_oracle_onc_align_strings_reproduce_golden(built):
    """onc_align WITH its gapped strings (what the consensus stage consumes): the oracle
    against vectors produced by the reference's own onc_align (tests/golden/make_golden_onc_align.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(util.GOLDEN, "make_golden_onc_align.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    gold = json.load(open(os.path.join(util.GOLDEN, "onc_align_a.json")))
    mine = mk.run("oracle")
    assert len(mine) == len(gold) == 2 * mk.N_CAND
    assert mine == gold
    assert sum(g["ok"] for g in gold) > 100 and any(g["tail"] == 4 for g in gold)


@pytest.mark.skipif(not os.path.exists(ora.REF_LIB), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_onc_align_matches_reference_on_fresh_pairs(built):
    """Random overlapping read pairs with anchors anywhere inside the overlap (left AND right extensions of
    several blocks, failed extensions, anchors at sequence ends), both tail lengths."""
    import numpy as np
    from necat_amd.synth import _mutate
    rng = np.random.default_rng(123)
    ao, ar = ora.Aligner(0.5, "oracle"), ora.Aligner(0.5, "ref")
    n = 0
    for it in range(60):
        g = rng.integers(0, 4, int(rng.integers(1500, 9000)), dtype=np.uint8)
        q = _mutate(g, float(rng.uniform(0.03, 0.16)), rng)
        t = _mutate(g, float(rng.uniform(0.03, 0.16)), rng)
        if it % 7 == 3:
            t = rng.integers(0, 4, t.shape[0], dtype=np.uint8)      # unrelated: the extension fails
        for _ in range(3):
            frac = float(rng.uniform(0.0, 1.0)) if it % 5 else float(rng.integers(0, 2))
            qs, ts = int(frac * (q.shape[0] - 1)), int(frac * (t.shape[0] - 1))
            for tail in (4, 1):
                assert ao.align(q, qs, t, ts, 500, tail) == ar.align(q, qs, t, ts, 500, tail), (it, qs, ts, tail)
                n += 1
    ao.close(); ar.close()
    assert n == 360


# ---- consensus stage, extension loop: oracle/cns_oracle.c vs the reference's own decisions ----

def _cns_cases():
    man = json.load(open(os.path.join(util.GOLDEN, "cns_c", "manifest.json")))
    return man


@pytest.mark.parametrize("case", ["default", "fixed", "cov6", "a2000"])
def test_cns_loop_golden(case, tmp_path):
    """every add_one_align call (target range, weight, both gapped strings) and the per-template numbers the
    reference logged (tests/golden/make_golden_cns.py) are reproduced by the restatement, byte for byte"""
    man = _cns_cases()
    wrk = util.install_golden_volumes(man["volumes"], tmp_path)
    for fn in ("cands.p0", "cands.partitions"):
        shutil.copy(os.path.join(util.GOLDEN, "cns_c", fn), os.path.join(str(tmp_path), fn))
    log = os.path.join(str(tmp_path), "ora.txt")
    ora.cns_run(ora.cns_options(**man["cases"][case]["options"]), wrk, os.path.join(str(tmp_path), "cands"), log)
    want = open(os.path.join(util.GOLDEN, "cns_c", "ref_%s.txt" % case)).read()
    got = open(log).read()
    assert got.count("\nT\t") + got.startswith("T\t") == man["cases"][case]["templates"]
    assert got == want


@pytest.mark.skipif(not ora.have_ref_cns(), reason="oracle/_ref (reference build) not present")
def test_cns_loop_fresh_vs_ref(tmp_path):
    """a fresh multi-volume dataset through reference oc2pmov -> oc2pcan -> consensus driver vs the restatement"""
    wrk, rs, nv = util.make_dataset(tmp_path, genome=30_000, coverage=30.0, seed=77, err=0.13, vol_size=300_000)
    assert nv > 1
    o = ora.options(**dict(util.FAST, job=0, binary_output=1, num_threads=2))
    can = os.path.join(str(tmp_path), "cands")
    with open(can, "wb") as f:
        for v in range(nv):
            ora.run_ref(o, v, wrk, can + ".v%d" % v)
            f.write(open(can + ".v%d" % v, "rb").read())
    ora.run_ref_pcan(wrk, can, batch_size=60)          # several partitions
    for kw in (dict(), dict(max_cov=8)):
        a, b = os.path.join(str(tmp_path), "ref.txt"), os.path.join(str(tmp_path), "ora.txt")
        ora.run_ref_cns(ora.cns_options(**kw), wrk, can, a, full=True)
        ora.cns_run(ora.cns_options(**kw), wrk, can, b, full=True)
        assert open(a).read() == open(b).read()
        assert len(ora.parse_cns_log(a)) > 50


@pytest.mark.skipif(not ora.have_ref_cns(), reason="oracle/_ref (reference build) not present")
def test_pcan_single_partition_vs_ref(tmp_path):
    """tests/util.py's role swap (used to build candidate partitions on the GPU box) = the reference's oc2pcan"""
    wrk = util.install_golden_volumes("vols_a", tmp_path)
    rec = open(os.path.join(util.GOLDEN, "a_fast_can_bin.bin"), "rb").read()
    can = os.path.join(str(tmp_path), "cands")
    open(can, "wb").write(rec)
    ora.run_ref_pcan(wrk, can)
    want = open(can + ".p0", "rb").read()
    got = util.pcan_single_partition(rec)
    key = lambda b: sorted(b[i:i + 28] for i in range(0, len(b), 28))
    assert len(want) == 2 * len(rec) and key(got) == key(want)


@pytest.mark.skipif(not ora.have_ref_cns(), reason="oracle/_ref (reference build) not present")
def test_cns_loop_deep_coverage_vs_ref(tmp_path):
    """more than MAX_EXAMINED_CAN = 300 candidates per template: the reference's sort + cut (consensus_one_read.c:250-260)
    and its loop over several groups of 50 vs the restatement"""
    wrk, rs, nv = util.make_dataset(tmp_path, genome=5_000, coverage=400.0, seed=23, err=0.12)
    o = ora.options(**dict(util.FAST, job=0, binary_output=1, num_threads=4))
    can = os.path.join(str(tmp_path), "cands")
    with open(can, "wb") as f:
        for v in range(nv):
            ora.run_ref(o, v, wrk, can + ".v%d" % v)
            f.write(open(can + ".v%d" % v, "rb").read())
    ora.run_ref_pcan(wrk, can)
    a, b = os.path.join(str(tmp_path), "ref.txt"), os.path.join(str(tmp_path), "ora.txt")
    kw = dict(max_cov=30)
    ora.run_ref_cns(ora.cns_options(**kw), wrk, can, a)
    ora.cns_run(ora.cns_options(**kw), wrk, can, b)
    assert open(a).read() == open(b).read()
    assert max(t[3] for t in ora.parse_cns_log(a)) > 50          # num_can: the loop went beyond the first group

