"""Full-size parity: BASELINE.json configs[1] (E. coli 4.6 Mb x 40, OVLP_FAST_OPTIONS) and configs[2] (S. cerevisiae-size
12 Mb x 50, OVLP_SENSITIVE_OPTIONS -z 10: 2.46 M candidates, several seeding chunks and extension batches at their real
sizes), k = 15.  The expected fingerprints were produced by the REFERENCE binary (oracle/_ref/oc2pmov, built from
/root/reference) on the same seeded datasets by tests/golden/make_golden_full.py and are committed as
tests/golden/{ecoli,yeast}_full_reference.json; nothing here needs /root/reference at run time."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
GOLDS = {n: json.load(open(os.path.join(util.GOLDEN, "%s_full_reference.json" % n))) for n in ("ecoli", "yeast")}


@pytest.fixture(scope="module", params=["ecoli", "yeast"])
def ecoli(ctx, request):
    from necat_amd import synth
    GOLD = GOLDS[request.param]
    g = GOLD["generator"]
    rs = synth.simulate_reads(g["genome"], g["coverage"], seed=g["seed"], err=g["err"])
    if hashlib.md5(rs.codes.tobytes()).hexdigest() != GOLD["reads_md5"]:
        pytest.skip("numpy generator drift: the seeded dataset differs from the one the golden was made on")
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
