"""Full-size parity: BASELINE.json configs[1] (E. coli 4.6 Mb x 40, OVLP_FAST_OPTIONS) and configs[2] (S. cerevisiae-size
12 Mb x 50, OVLP_SENSITIVE_OPTIONS -z 10: 2.46 M candidates, several seeding chunks and extension batches at their real
sizes), k = 15.  The expected fingerprints were produced by the REFERENCE binary (oracle/_ref/oc2pmov, built from
/root/reference) on the same seeded datasets by tests/golden/make_golden_full.py and are committed as
tests/golden/{ecoli,yeast}_full_reference.json; nothing here needs /root/reference at run time.

Round 6 - SURVEY 8d's other data shapes at configs[1]'s size, same generator script, same reference binary: "ecoli_repeats" (29 % of the genome in three repeat
families: 15-mers above the -q 500 cutoff, lookup_table.c:15-58; blocks at their 40 seeds, word_finder.c:91-92; reads with more than -n 500 candidates,
pm_worker.c:139-140, :168-171), "ecoli_err6" (6 % errors) and "ecoli_longtail" (log-normal read lengths up to 180 kb: chains of 300 blocks)."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
SHAPES = ("ecoli", "yeast", "ecoli_repeats", "ecoli_err6", "ecoli_longtail")
GOLDS = {n: json.load(open(os.path.join(util.GOLDEN, "%s_full_reference.json" % n))) for n in SHAPES}


def _generate(g):
    from necat_amd import synth
    return synth.simulate_reads(g["genome"], g["coverage"], seed=g["seed"], err=g["err"], families=g.get("families"), lognormal=g.get("lognormal"), max_len=g.get("max_len"))


@pytest.fixture(scope="module", params=list(SHAPES))
def ecoli(ctx, request):
    from necat_amd import synth
    GOLD = GOLDS[request.param]
    g = GOLD["generator"]
    rs = _generate(g)
    # (a FAILURE, not a skip: a numpy whose generators drifted would otherwise silently turn the strongest parity tests off)
    assert hashlib.md5(rs.codes.tobytes()).hexdigest() == GOLD["reads_md5"], \
        "numpy generator drift: the seeded dataset differs from the one tests/golden/%s_full_reference.json was made on - regenerate the golden (make_golden_full.py)" % request.param
    vol = ctx.upload_volume(synth.pack_2bit(rs.codes), rs.nbases, rs.offsets, rs.sizes)
    ix = ctx.build_index(vol, 15, 500)
    yield rs, vol, ix, GOLD
    ix.free()
    vol.free()


def _opt(job, gold):
    from necat_amd import capi
    flags = gold["options"].split()
    return capi.default_options(kmer_size=15, scan_window=int(flags[flags.index("-z") + 1]), kmer_cnt_cutoff=500, block_size=2000, block_score_cutoff=3,
                                num_candidates=500, align_size_cutoff=1000, error=0.5, job=job, use_hdr_as_id=0)


def test_candidates_identical_to_reference(ctx, ecoli):
    from necat_amd import capi
    rs, vol, ix, GOLD = ecoli
    c = ctx.find_candidates(ix, vol, vol, 0, 0, _opt(0, GOLD), True)
    recs = sorted(bytes(r) for r in capi.pack_candidates(c).astype("<u4"))
    assert len(recs) == GOLD["candidate_records"]
    assert hashlib.md5(b"".join(recs)).hexdigest() == GOLD["candidates_packed_sorted_md5"]


def test_m4_identical_to_reference(ctx, ecoli):
    from necat_amd import capi
    rs, vol, ix, GOLD = ecoli
    opt = _opt(1, GOLD)
    c = ctx.find_candidates(ix, vol, vol, 0, 0, opt, True)
    m4 = ctx.extend(vol, vol, 0, 0, c, opt, 1)
    assert m4.shape[0] == GOLD["m4_records"]
    assert int((m4["qend"] - m4["qoff"]).sum()) == GOLD["aligned_query_bases"]
    lines = sorted(capi.m4_text_lines(m4))
    assert hashlib.md5(b"".join(lines)).hexdigest() == GOLD["m4_text_sorted_md5"]
    # size-independent properties of the records
    assert np.all(m4["qoff"] < m4["qend"]) and np.all(m4["qend"] <= m4["qsize"])
    assert np.all(m4["soff"] < m4["send"]) and np.all(m4["send"] <= m4["ssize"])
    assert np.all(m4["sid"] < m4["qid"])            # self-volume: only subjects before the query (word_finder.c:121-127)
    assert np.all((m4["ident_perc"] > 50.0) & (m4["ident_perc"] <= 100.0))


def test_m4_identical_with_the_batch_cut_in_two(ecoli, monkeypatch):
    """the same records when the one batch of candidates (E. coli: 228 k; yeast runs its four batches on two lanes by default) is cut in two whose rounds run side
    by side - NECAT_EXT_OVERLAP_MIN, knobs.h; a context of its own: the knobs are read when a context is created"""
    from necat_amd import capi
    rs, vol, ix, GOLD = ecoli
    monkeypatch.setenv("NECAT_EXT_OVERLAP_MIN", "100000")
    monkeypatch.setenv("NECAT_EXT_OVERLAP_SPLIT", "45" if GOLD is GOLDS["yeast"] else "20")
    monkeypatch.setenv("NECAT_BATCH", "3000000")           # (yeast: its 2.46 M candidates as ONE batch, cut 45 : 55)
    c = capi.Context(0)
    try:
        from necat_amd import synth
        v = c.upload_volume(synth.pack_2bit(rs.codes), rs.nbases, rs.offsets, rs.sizes)
        x = c.build_index(v, 15, 500)
        m4, _ = c.map_pair(x, v, v, 0, 0, _opt(1, GOLD), True, 1)
        x.free(); v.free()
    finally:
        c.close()
    assert m4.shape[0] == GOLD["m4_records"]
    assert hashlib.md5(b"".join(sorted(capi.m4_text_lines(m4)))).hexdigest() == GOLD["m4_text_sorted_md5"]


# ---- multi-volume projects at real volume sizes through the oc2pm PROGRAM: "multivol" = 1.48 Gbp in three volumes of 1.05 / 0.30 / 0.13 Gbp;
# "drosophila" = BASELINE configs[3] at its real size, a 140 Mb genome x 40 = 5.6 Gbp cut by oc2mkdb's own 2 Gbp rule into 2.0 / 2.0 / 1.6 Gbp
# (the volume size at which 34-bit offsets, u32 slot counts and the 786 432-candidate batch cap are real; 6.9 M records per mode);
# "human_subset" = BASELINE configs[4] as a stated subset: a 3 Gb genome read at the 30x rate, the first three of its 45 oc2mkdb volumes (2.0 / 2.0 / 2.0 Gbp =
# 2x of the genome): ~ 2 x 10^9 k-mer positions per volume that are nearly all distinct (table occupancy ~ 0.85 of 4^15, the -q cut idle), a handful of true
# overlaps per read instead of ~ 80 - index build and seeding at a hit density no smaller genome has (word_finder.c:121-127, lookup_table.h:12-15)
MV_SETS = {}
for _name in ("multivol", "drosophila", "human_subset"):
    _p = os.path.join(util.GOLDEN, "%s_full_reference.json" % _name)
    if os.path.exists(_p):
        MV_SETS[_name] = json.load(open(_p))


@pytest.fixture(scope="module", params=sorted(MV_SETS))
def multivol_dir(tmp_path_factory, request):
    """the seeded read set of the golden, cut into its volumes (synth.write_volume_dir_cuts; for `drosophila` the cuts are oc2mkdb's 2 Gbp)"""
    from necat_amd import synth
    MV = MV_SETS[request.param]
    g = MV["generator"]
    if MV["nbases"] > 3_000_000_000:
        try:
            import psutil
            if psutil.virtual_memory().available < 48 << 30:
                pytest.skip("the 5.6 Gbp read set needs ~ 48 GB of host memory to generate")
        except ImportError:
            pass
    rs = synth.simulate_reads(g["genome"], g["coverage"], seed=g["seed"], err=g["err"])
    assert hashlib.md5(rs.codes.tobytes()).hexdigest() == MV["reads_md5"], \
        "numpy generator drift: the seeded dataset differs from the one tests/golden/%s_full_reference.json was made on - regenerate the golden (make_golden_multivol.py)" % request.param
    keep = os.environ.get("NECAT_TEST_KEEP_VOLS")          # tools/r04: the profile pass reuses the volumes this fixture wrote
    d = os.path.join(keep, request.param) if keep else os.path.join(str(tmp_path_factory.mktemp("mv")), "vols")
    assert synth.write_volume_dir_cuts(d, rs, g["cuts"]) == MV["volumes"]
    del rs
    yield d, MV
    if not keep:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("mode", ["can", "m4"])
def test_multivolume_project_through_oc2pm_equals_the_reference(multivol_dir, built, mode):
    """all six (reference volume, query volume) pairs, two oc2pm workers on the device splitting them by the pair schedule
    (pair_sched.h): the concatenated output holds exactly the records the REFERENCE's three oc2pmov jobs wrote
    (tests/golden/make_golden_multivol.py: sorted md5 over all volumes)"""
    import subprocess
    pmov, pm = built.build_cli()
    d, MV = multivol_dir
    for f in os.listdir(d):
        if f.startswith("pm") and f.endswith(".finished"):
            os.remove(os.path.join(d, f))
    out = os.path.join(d, "all_" + mode)
    extra = ["-j", "0", "-u", "1", "-i", "1"] if mode == "can" else ["-j", "1", "-u", "0", "-i", "0"]
    env = dict(os.environ, NECAT_GPUS="0,0")
    r = subprocess.run([pm] + MV["options"].split() + extra + ["-t", "8", d, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("unit(s) of job") >= 3          # both workers ran units, volume 0's pairs were split between them
    if mode == "can":
        raw = np.fromfile(out, dtype="<u4").reshape(-1, 7)
        os.remove(out)
        assert raw.shape[0] == MV["candidate_records"]
        # sorted as 28-byte strings (the golden sorted bytes objects): a lexicographic sort over the records' byte columns
        order = np.lexsort(raw.view(np.uint8).reshape(-1, 28).T[::-1])
        assert hashlib.md5(raw[order].tobytes()).hexdigest() == MV["candidates_packed_sorted_md5"]
    else:
        lines = sorted(open(out, "rb").read().splitlines(keepends=True))
        os.remove(out)
        assert len(lines) == MV["m4_records"]
        assert hashlib.md5(b"".join(lines)).hexdigest() == MV["m4_text_sorted_md5"]
