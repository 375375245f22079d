"""oc2rm_worker without a GPU: the host side of necat_map_reference (necat_amd/csrc/rm_host.h: containment, drop, rescue pair on
the stretch of the reference a read can reach, rm_window) behind the oracle's seeding and block-wise aligner
(tests/host_core/check_rm.cpp), against the output of the REFERENCE's own oc2rm_worker -t 1 (oracle/_ref, built from
/root/reference)."""
import os
import subprocess

import pytest

from oracle import oracle_api as ora
from tests import util

needs_ref = pytest.mark.skipif(not os.path.exists(ora.REF_RM), reason="oracle/_ref/oc2rm_worker (the reference's build) is absent")


@pytest.fixture(scope="module")
def check_rm(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("rm"))
    obj = os.path.join(d, "necat_oracle.o")
    subprocess.run(["gcc", "-O2", "-std=gnu99", "-c", os.path.join(util.ROOT, "oracle", "necat_oracle.c"), "-o", obj], check=True)
    exe = os.path.join(d, "check_rm")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(util.ROOT, "tests", "host_core", "check_rm.cpp"),
                    obj, "-lm", "-lpthread"], check=True)
    return exe


def test_rm_golden(check_rm, tmp_path):
    """the committed vectors (tests/golden/rm_e: what the reference's oc2rm_worker wrote for tests/golden/vols_e against rm_e/ref.vol)"""
    import json
    m = json.load(open(os.path.join(util.GOLDEN, "manifest_asm_rm.json")))["rm_e"]
    wrk = util.install_golden_volumes(m["volumes"], tmp_path)
    got = os.path.join(str(tmp_path), "mine.m4")
    r = subprocess.run([check_rm] + m["args"].split() + [wrk, os.path.join(util.GOLDEN, "rm_e", "ref.vol"), got], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    want = open(os.path.join(util.GOLDEN, "rm_e", "ref.m4")).read()
    assert open(got).read() == want
    assert len(want.splitlines()) == m["records"] and int(dict(kv.split("=") for kv in r.stdout.split())["rescued"]) > 3


@needs_ref
@pytest.mark.parametrize("seed,repeat,args", [
    (13, 0.6, "-k 13 -i 0"),
    (12, 0.4, "-k 12 -z 10 -n 8 -a 1000 -i 0"),
    (17, 0.2, "-k 13 -b 2000 -e 0.3 -i 0"),
])
def test_rm_replay_matches_reference(check_rm, tmp_path, seed, repeat, args):
    wrk, ref, nv = util.make_rm_dataset(tmp_path, seed=seed, repeat_frac=repeat)
    want, got = os.path.join(str(tmp_path), "ref.m4"), os.path.join(str(tmp_path), "mine.m4")
    subprocess.run([ora.REF_RM] + args.split() + ["-t", "1", wrk, ref, want], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    r = subprocess.run([check_rm] + args.split() + [wrk, ref, got], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    f = dict(kv.split("=") for kv in r.stdout.split())
    assert int(f["records"]) > 100 and int(f["rescued"]) > 5
    assert open(got).read() == open(want).read()
