"""GPU parity of the consensus stage's extension loop: necat_cns_extension_batch through the C ABI
vs (a) what the REFERENCE decided on the golden partition (tests/golden/cns_c, logged from the reference's own
consensus driver) and (b) the oracle's sequential restatement on fresh data.  Compared as text in the log format
of oracle/cns_ref_harness.c: every add_one_align call (target range, weight, length, both gapped strings) in call
order + per template ident_cutoff, num_can, num_ovlps, cov_ranges."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from necat_amd import capi
from oracle import oracle_api as ora
from tests import util

pytestmark = pytest.mark.gpu


def _run(ctx, wrk, partition_bytes, okw, full=False):
    vol = ctx.load_merged_volumes(wrk)
    cands, off, n_all = ctx.cns_load_partition(vol, np.frombuffer(partition_bytes, dtype=np.uint8))
    res = ctx.cns_extension_batch(vol, cands, off, n_all, capi.cns_options(**okw))
    roff = np.zeros(len(vol.names) + 1, dtype=np.int64)
    roff[1:] = np.cumsum(vol.sizes)
    txt = util.cns_log_text(res, cands, off, vol.codes, roff, ora.fnv64, full=full)
    stats = (res.n_aligned, res.n_used, res.n_rounds)
    res.free()
    vol.free()
    return txt, stats


@pytest.mark.parametrize("case", ["default", "fixed", "cov6", "a2000"])
def test_cns_loop_golden(ctx, tmp_path, case):
    man = json.load(open(os.path.join(util.GOLDEN, "cns_c", "manifest.json")))
    wrk = util.install_golden_volumes(man["volumes"], tmp_path)
    part = open(os.path.join(util.GOLDEN, "cns_c", "cands.p0"), "rb").read()
    txt, stats = _run(ctx, wrk, part, man["cases"][case]["options"])
    want = open(os.path.join(util.GOLDEN, "cns_c", "ref_%s.txt" % case)).read()
    assert txt.count("\nT\t") + txt.startswith("T\t") == man["cases"][case]["templates"]
    assert txt == want
    assert stats[1] <= stats[0]


@pytest.mark.parametrize("okw,knobs", [
    (dict(), {}),
    (dict(max_cov=8, min_cov=2), {"NECAT_CNS_SPEC": "1", "NECAT_CNS_SPEC_EXTRA": "-1"}),     # no speculation at all
    (dict(use_fixed_ident_cutoff=1, error=0.3), {"NECAT_CNS_SPEC": "40", "NECAT_CNS_SPEC_EXTRA": "35"}),
    (dict(), {"NECAT_BATCH": "1024"}),                                                        # passes of several batches
])
def test_cns_loop_fresh_vs_oracle(built, tmp_path, okw, knobs):
    wrk, rs, nv = util.make_dataset(tmp_path, genome=40_000, coverage=35.0, seed=91, err=0.13, vol_size=500_000)
    o = ora.options(**dict(util.FAST, job=0, binary_output=1, num_threads=4))
    rec = b""
    for v in range(nv):
        out = os.path.join(str(tmp_path), "pm_%d" % v)
        ora.pm_main(o, v, wrk, out)
        rec += open(out, "rb").read()
    part = util.pcan_single_partition(rec)
    prefix = os.path.join(str(tmp_path), "cands")
    util.write_partition(prefix, part)
    log = os.path.join(str(tmp_path), "ora.txt")
    ora.cns_run(ora.cns_options(**okw), wrk, prefix, log, full=True)
    old = {k: os.environ.get(k) for k in knobs}
    os.environ.update(knobs)
    try:
        c = capi.Context(0)        # knobs are read when a context is created
        txt, stats = _run(c, wrk, part, okw, full=True)
        c.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    want = open(log).read()
    assert want.count("\nT\t") > 100
    assert txt == want
    if knobs.get("NECAT_CNS_SPEC") == "1":
        assert stats[0] == stats[1]            # nothing speculative was computed


def test_cns_argument_errors(ctx, tmp_path):
    """inconsistent input is refused with NECAT_ERR_ARG and a message, nothing is computed"""
    man = json.load(open(os.path.join(util.GOLDEN, "cns_c", "manifest.json")))
    wrk = util.install_golden_volumes(man["volumes"], tmp_path)
    part = np.frombuffer(open(os.path.join(util.GOLDEN, "cns_c", "cands.p0"), "rb").read(), dtype=np.uint8)
    vol = ctx.load_merged_volumes(wrk)
    cands, off, n_all = ctx.cns_load_partition(vol, part)
    co = capi.cns_options()
    for field, value in (("sdir", 1), ("sid", int(cands[0]["sid"]) + 1), ("send", int(cands[0]["ssize"]) + 1), ("qsize", 17)):
        bad = cands.copy()
        bad[field][3] = value
        with pytest.raises(capi.NecatError, match="inconsistent"):
            ctx.cns_extension_batch(vol, bad, off, n_all, co)
    with pytest.raises(capi.NecatError, match="out of range"):
        ctx.cns_extension_batch(vol, cands, off, n_all, capi.cns_options(max_cov=0))
    rec = part.copy().view("<u4").reshape(-1, 7)
    rec[5, 1] = 10 ** 6                                    # a template id outside the read set
    with pytest.raises(capi.NecatError, match="outside the read set"):
        ctx.cns_load_partition(vol, rec.view(np.uint8).reshape(-1))
    # an empty call and a call whose templates all have fewer than min_cov candidates are fine
    r = ctx.cns_extension_batch(vol, cands[:0], np.zeros(1, dtype=np.uint64), None, co)
    assert r.templates.shape[0] == 0 and r.overlaps.shape[0] == 0
    r.free()
    r = ctx.cns_extension_batch(vol, cands, off, n_all, capi.cns_options(min_cov=10 ** 6))
    assert r.templates.shape[0] == off.shape[0] - 1 and not r.templates["examined"].any() and r.n_aligned == 0
    r.free()
    vol.free()


def test_front_half_chain_reads_to_overlaps(ctx, built, tmp_path):
    """The whole front half of the correction pipeline with this repo's programs only - FASTA -> oc2mkdb -> oc2pmov
    -j 0 -u 1 per volume (GPU) -> oc2pcan -> necat_cns_extension_batch (GPU) - must hand the consensus exactly the
    overlaps the REFERENCE chain (its oc2mkdb, oc2pmov, oc2pcan, consensus driver) produced for the same reads:
    tests/golden/cns_c/ref_default.txt."""
    import subprocess
    from necat_amd import build, synth
    pmov, _ = built.build_cli()
    man = json.load(open(os.path.join(util.GOLDEN, "cns_c", "manifest.json")))
    src = util.install_golden_volumes(man["volumes"], os.path.join(str(tmp_path), "src"))
    fa = os.path.join(str(tmp_path), "reads.fasta")
    with open(fa, "w") as f:
        for path, _, _ in capi.load_volumes_info(src)[2]:
            pac, off, sz, names = synth.read_volume(path)
            codes = synth.unpack_2bit(pac, int(sz.sum()))
            for i, nm in enumerate(names):
                f.write(">%s\n%s\n" % (nm, bytes(b"ACGT"[c] for c in codes[int(off[i]):int(off[i] + sz[i])]).decode()))
    lst = os.path.join(str(tmp_path), "list.txt")
    open(lst, "w").write(fa + "\n")
    wrk = os.path.join(str(tmp_path), "wrk")
    env = dict(os.environ, NECAT_MKDB_VOLSIZE="500000")           # the golden set was cut into volumes of >= 500 kbp
    assert subprocess.run([build.OC2MKDB, wrk, lst], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).returncode == 0
    nv, nr, _ = capi.load_volumes_info(wrk)
    assert nv == man["n_volumes"]
    o = ora.options(**dict(util.FAST, job=0, binary_output=1, num_threads=1))
    can = os.path.join(str(tmp_path), "cands")
    with open(can, "wb") as f:
        for v in range(nv):
            out = os.path.join(str(tmp_path), "pm_result_%d" % v)
            r = subprocess.run([pmov] + ora.opt_argv(o) + [wrk, str(v), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            assert r.returncode == 0, r.stderr
            f.write(open(out, "rb").read())
    assert os.path.getsize(can) // 28 == man["candidates"]
    assert subprocess.run([build.OC2PCAN, wrk, can], stdout=subprocess.PIPE, stderr=subprocess.STDOUT).returncode == 0
    assert open(can + ".partitions").read() == "1\n"
    txt, stats = _run(ctx, wrk, open(can + ".p0", "rb").read(), {})
    assert txt == open(os.path.join(util.GOLDEN, "cns_c", "ref_default.txt")).read()



# ---- the oc2cns PROGRAM (necat_amd/csrc/oc2cns: GPU extension loop + host consensus) against the reference's oc2cns ----

def _run_oc2cns(built, argv, wrk, can, tmp, tag, mn=None):
    built.build_cli()
    co, ro = os.path.join(tmp, "cns_" + tag), os.path.join(tmp, "raw_" + tag)
    cmd = [built.OC2CNS] + argv + [wrk, can, co, ro] + (["-mn", str(mn[0]), str(mn[1])] if mn else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    return open(co, "rb").read(), open(ro, "rb").read()


@pytest.mark.parametrize("name", sorted(json.load(open(os.path.join(util.GOLDEN, "cns_c", "manifest.json")))["oc2cns"]))
def test_oc2cns_program_reproduces_reference_files(built, tmp_path, name):
    """both output files byte-identical to what the reference's oc2cns -t 1 wrote for the golden partition"""
    import hashlib
    m = json.load(open(os.path.join(util.GOLDEN, "cns_c", "manifest.json")))["oc2cns"][name]
    wrk = util.install_golden_volumes("vols_c", tmp_path)
    can = os.path.join(str(tmp_path), "cands")
    for fn in ("cands.p0", "cands.partitions"):
        shutil.copy(os.path.join(util.GOLDEN, "cns_c", fn), os.path.join(str(tmp_path), fn))
    cns, raw = _run_oc2cns(built, ora.cns_argv(ora.cns_options(**m["options"])) + m["extra_argv"] + ["-t", "3"], wrk, can, str(tmp_path), name)
    assert (len(cns), cns.count(b">")) == (m["cns_bytes"], m["cns_records"]) and hashlib.md5(cns).hexdigest() == m["cns_md5"]
    assert (len(raw), raw.count(b">")) == (m["raw_bytes"], m["raw_records"]) and hashlib.md5(raw).hexdigest() == m["raw_md5"]


@pytest.mark.skipif(not os.path.exists(ora.REF_OC2CNS), reason="needs oracle/_ref (built from /root/reference; it travels to the GPU box)")
def test_oc2cns_program_sparse_partitions_and_nodes(built, tmp_path):
    """sparse coverage (raw intervals, uncorrected reads), several partitions, and the -mn node split: node 0 + node 1 together
    write what one run writes, and that is what the reference writes"""
    d, rs, nv = util.make_dataset(tmp_path, genome=80_000, coverage=7.0, seed=77, err=0.12, vol_size=300_000)
    can = util.reference_candidate_partitions(d, nv, str(tmp_path))
    # re-partition into several files (reference oc2pcan, small batch size) so that -mn has something to split
    subprocess.run([ora.REF_PCAN, "-p", "20", "-t", "1", d, can], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    nparts = int(open(can + ".partitions").read().split()[0])
    assert nparts >= 2
    o = ora.cns_options(min_cov=3)
    argv = ora.cns_argv(o)
    rc, rr = os.path.join(str(tmp_path), "ref_cns"), os.path.join(str(tmp_path), "ref_raw")
    subprocess.run([ora.REF_OC2CNS] + argv + ["-t", "1", d, can, rc, rr], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    cns, raw = _run_oc2cns(built, argv + ["-t", "2"], d, can, str(tmp_path), "all")
    assert cns == open(rc, "rb").read() and cns.count(b">") > 40
    assert raw == open(rr, "rb").read() and raw.count(b">") > 0
    parts = [_run_oc2cns(built, argv, d, can, str(tmp_path), "n%d" % n, mn=(n, 2)) for n in range(2)]
    recs = lambda b: sorted(b.split(b">"))
    assert recs(parts[0][0] + parts[1][0]) == recs(cns) and recs(parts[0][1] + parts[1][1]) == recs(raw)
    # -s 1 (SMALL_MEMORY pipelines): the reference never loads `reads` then, so the partition-end dump of the reads nobody corrected
    # (consensus_one_partition.c:172-194, inside `if (reads)`) does not happen: raw_out is shorter, cns_out the same
    subprocess.run([ora.REF_OC2CNS] + argv + ["-s", "1", "-t", "1", d, can, rc + "_s1", rr + "_s1"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    cns_s, raw_s = _run_oc2cns(built, argv + ["-s", "1", "-t", "2"], d, can, str(tmp_path), "s1")
    assert cns_s == open(rc + "_s1", "rb").read() == cns
    assert raw_s == open(rr + "_s1", "rb").read() and len(raw_s) < len(raw)


# ---- -r 1: the host rescue pair behind the device pass (cns_rescue.h) ----

@pytest.mark.skipif(not (ora.have_ref_cns() and os.path.exists(ora.REF_OC2CNS)),
                    reason="needs oracle/_ref (built from /root/reference; it travels to the GPU box)")
def test_cns_rescue_long_indels_vs_reference(ctx, built, tmp_path):
    """reads with long indels, rescue_long_indels = 1: the loop's decisions and alignments through the C ABI against the log of the
    REFERENCE's consensus driver run with -r 1, then the oc2cns program's two files against the reference's oc2cns -r 1"""
    wrk, can, part = util.make_long_indel_partition(tmp_path)
    want = os.path.join(str(tmp_path), "ref_r1.txt")
    subprocess.run([ora.REF_CNS] + ora.cns_argv(ora.cns_options()) + ["-r", "1", wrk, can, want, "full"], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    vol = ctx.load_merged_volumes(wrk)
    cands, off, n_all = ctx.cns_load_partition(vol, np.frombuffer(part, dtype=np.uint8))
    res = ctx.cns_extension_batch(vol, cands, off, n_all, capi.cns_options(rescue_long_indels=1))
    roff = np.zeros(len(vol.names) + 1, dtype=np.int64)
    roff[1:] = np.cumsum(vol.sizes)
    txt = util.cns_log_text(res, cands, off, vol.codes, roff, ora.fnv64, full=True)
    assert res.n_rescued > 150 and res.n_rescue_tried >= res.n_rescued
    res.free()
    res0 = ctx.cns_extension_batch(vol, cands, off, n_all, capi.cns_options())
    assert res0.n_rescue_tried == 0 and res0.n_rescued == 0
    txt0 = util.cns_log_text(res0, cands, off, vol.codes, roff, ora.fnv64, full=True)
    res0.free()
    vol.free()
    assert txt == open(want).read()
    assert txt0 != txt
    argv = ora.cns_argv(ora.cns_options()) + ["-r", "1"]
    rc, rr = os.path.join(str(tmp_path), "ref_cns"), os.path.join(str(tmp_path), "ref_raw")
    subprocess.run([ora.REF_OC2CNS] + argv + ["-t", "1", wrk, can, rc, rr], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    cns, raw = _run_oc2cns(built, argv + ["-t", "3"], wrk, can, str(tmp_path), "r1")
    assert cns == open(rc, "rb").read() and cns.count(b">") > 40
    assert raw == open(rr, "rb").read()
