"""CPU test of the kernels' per-lane cores: necat_amd/csrc/{seed,dp,ext}_core.h are compiled with g++
(tests/host_core/check_core.cpp) and replayed lane by lane against the oracle - every candidate of
every read, every block alignment (distance, end column, traceback, tail trimming, identity)."""
import os
import subprocess

import pytest

from tests import util


@pytest.fixture(scope="module")
def check_core(tmp_path_factory, built):
    d = tmp_path_factory.mktemp("hc")
    exe = os.path.join(str(d), "check_core")
    obj = os.path.join(str(d), "necat_oracle.o")
    subprocess.run(["gcc", "-O2", "-std=gnu99", "-c", os.path.join(util.ROOT, "oracle", "necat_oracle.c"), "-o", obj], check=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                    os.path.join(util.ROOT, "tests", "host_core", "check_core.cpp"), obj, "-lm", "-lpthread"], check=True)
    return exe


@pytest.mark.parametrize("ds,vid,args", [
    ("vols_a", 0, "13 20 500 2000 3 500 1000 0.5".split()),
    ("vols_b", 0, "12 10 100 2000 3 500 1000 0.5".split()),
    ("vols_b", 1, "12 10 100 2000 3 3 400 0.5".split()),
    ("vols_a", 0, "11 5 500 1000 3 20 400 0.5 100000 0".split()),
])
def test_cores_match_oracle(check_core, tmp_path, ds, vid, args):
    d = util.install_golden_volumes(ds, tmp_path)
    r = subprocess.run([check_core, d, str(vid)] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "seed_mismatch=0 ext_mismatch=0" in r.stdout
    assert "candidates=0 " not in r.stdout
