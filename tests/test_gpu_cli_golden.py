"""GPU tests of the process boundary: the oc2pmov / oc2pm programs, run exactly as necat.pl runs them
(necat.pl:197), must write record files that are byte-identical - after sorting records - to what the
REFERENCE wrote for the same volumes and flags (tests/golden, generated from oracle/_ref)."""
import hashlib
import json
import os
import subprocess

import pytest

from tests import util
from oracle import oracle_api as ora

pytestmark = pytest.mark.gpu
MANIFEST = json.load(open(os.path.join(util.GOLDEN, "manifest.json")))


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_oc2pmov_reproduces_reference_records(name, tmp_path, built):
    pmov, _ = built.build_cli()
    m = MANIFEST[name]
    d = util.install_golden_volumes(m["dataset"], tmp_path)
    o = ora.options(**m["options"])
    out = os.path.join(str(tmp_path), "pm_result")
    r = subprocess.run([pmov] + ora.opt_argv(o) + [d, str(m["vid"]), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert "'pairwise mapping v%d vs v%d' takes" % (m["vid"], m["vid"]) in r.stdout
    recs = ora.sorted_records(out, ora.record_size(o))
    assert len(recs) == m["records"]
    assert hashlib.md5(b"".join(recs)).hexdigest() == m["md5"]
    assert not os.path.exists(out + ".part")


@pytest.mark.parametrize("gpus", [None, "0,0"])
def test_oc2pm_wrapper_concatenates_volumes(tmp_path, built, gpus):
    """gpus = "0,0": two oc2pmov children at a time (one per listed device - here the same one twice),
    the multi-GPU scheduling of reference volumes."""
    pmov, pm = built.build_cli()
    d = util.install_golden_volumes("vols_b", tmp_path)
    o = ora.options(**MANIFEST["b_v0_m4_txt"]["options"])
    out = os.path.join(str(tmp_path), "all.m4")
    env = dict(os.environ)
    if gpus:
        env["NECAT_GPUS"] = gpus
    r = subprocess.run([pm] + ora.opt_argv(o) + [d, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert r.returncode == 0, r.stderr
    got = sorted(open(out, "rb").read().splitlines(keepends=True))
    want = []
    for vid in range(3):
        p = os.path.join(str(tmp_path), "o%d" % vid)
        ora.pm_main(o, vid, d, p)
        want += open(p, "rb").read().splitlines(keepends=True)
    assert got == sorted(want)
    for vid in range(3):       # pairwise_mapping/main.c:55-70, :104-112
        assert os.path.exists(os.path.join(d, "pm%d.finished" % vid))
        assert not os.path.exists(os.path.join(d, "pm_result_%d" % vid))


def test_oc2pm_worker_failure_is_reported(tmp_path, built):
    """a volume job that fails (here: an unreadable volume file) makes oc2pm exit 1, leaves no pm<i>.finished for that volume and no
    merged output - the grid driver re-queues the script (pairwise_mapping/main.c:95-112 checks the child's exit status)"""
    pmov, pm = built.build_cli()
    d = util.install_golden_volumes("vols_b", tmp_path)
    o = ora.options(**MANIFEST["b_v0_m4_txt"]["options"])
    nv, nr, vols = __import__("necat_amd.capi", fromlist=["x"]).load_volumes_info(d)
    os.truncate(vols[2][0], 40)                     # volume 2 is needed by every job
    out = os.path.join(str(tmp_path), "all.m4")
    r = subprocess.run([pm] + ora.opt_argv(o) + [d, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1
    assert "ERROR" in r.stderr
    assert not os.path.exists(out)
    assert not any(os.path.exists(os.path.join(d, "pm%d.finished" % v)) for v in range(3))
