"""oc2asmpm without a GPU: necat_amd/csrc/asm_core.h - block vote, MEM chain, 2048-bp block extension with DALIGNER
end extension - behind the oracle's lookup table and block aligner (tests/host_core/check_asmpm.cpp), against the output of the
REFERENCE's own oc2asmpm -t 1 (oracle/_ref, built from /root/reference) on corrected-read-like data: byte-identical files."""
import os
import subprocess

import pytest

from necat_amd import synth
from oracle import oracle_api as ora
from tests import util

REF_ASMPM = os.path.join(os.path.dirname(ora.REF_PMOV), "oc2asmpm")
needs_ref = pytest.mark.skipif(not os.path.exists(REF_ASMPM), reason="oracle/_ref/oc2asmpm (the reference's build) is absent")


@pytest.fixture(scope="module")
def check_asmpm(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("asmpm"))
    obj = os.path.join(d, "necat_oracle.o")
    subprocess.run(["gcc", "-O2", "-std=gnu99", "-c", os.path.join(util.ROOT, "oracle", "necat_oracle.c"), "-o", obj], check=True)
    exe = os.path.join(d, "check_asmpm")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(util.ROOT, "tests", "host_core", "check_asmpm.cpp"),
                    obj, "-lm", "-lpthread"], check=True)
    return exe


def test_asmpm_golden(check_asmpm, tmp_path):
    """the committed vectors (tests/golden/asm_d: what the reference's oc2asmpm wrote for tests/golden/vols_d; make_golden_asm_rm.py)"""
    import json
    m = json.load(open(os.path.join(util.GOLDEN, "manifest_asm_rm.json")))["asm_d"]
    wrk = util.install_golden_volumes(m["volumes"], tmp_path)
    n = 0
    for v in range(m["n_volumes"]):
        got = os.path.join(str(tmp_path), "mine_%d.m4" % v)
        r = subprocess.run([check_asmpm] + m["args"].split() + [wrk, str(v), got], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        want = open(os.path.join(util.GOLDEN, "asm_d", "ref_v%d.m4" % v), "rb").read()
        assert open(got, "rb").read() == want, v
        n += len(want.splitlines())
    assert n == m["records"] and n > 300


@needs_ref
@pytest.mark.parametrize("seed,err,repeat,indels,args", [
    (41, 0.03, 0.3, False, "-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400 -k 13"),      # necat.pl:36 (ASM_OVLP_OPTIONS)
    (42, 0.06, 0.5, False, "-z 5 -k 12 -n 20 -u 0"),
    (43, 0.01, 0.2, True, "-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400 -k 13"),
])
def test_asmpm_matches_reference(check_asmpm, tmp_path, seed, err, repeat, indels, args):
    rs = synth.simulate_reads(60_000, 12.0, seed=seed, err=err, repeat_frac=repeat)
    if indels:
        rs = synth.add_long_indels(rs, 0.3, seed=seed + 1)
    wrk = os.path.join(str(tmp_path), "vols")
    nv = synth.write_volume_dir(wrk, rs, 300_000)
    assert nv >= 2
    total = 0
    for v in range(nv):
        want, got = os.path.join(str(tmp_path), "ref_%d.m4" % v), os.path.join(str(tmp_path), "mine_%d.m4" % v)
        subprocess.run([REF_ASMPM] + args.split() + ["-t", "1", wrk, str(v), want], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        a = open(want, "rb").read()
        for batch in ("0", "1", "2"):   # candidate by candidate; the two-phase walk of the program (all anchors planned, aligned, finished); the same finished on packed columns
            r = subprocess.run([check_asmpm] + args.split() + [wrk, str(v), got], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                               env=dict(os.environ, CHECK_ASM_BATCH=batch))
            assert r.returncode == 0, r.stdout
            assert open(got, "rb").read() == a, (v, batch)
        total += len(a.splitlines())
    assert total > 1000
