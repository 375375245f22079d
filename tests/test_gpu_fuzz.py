"""GPU parity on random datasets x random option sets (k, z, q, b, s, n, a, e, 1-3 volumes, repeats): the
sweep of tests/tools/fuzz_parity.py, a few cases per run.  Candidates (-j 0) and M4 records (-j 1) must equal the
oracle's for every volume."""
import os
import subprocess
import sys

import pytest

from tests import util

pytestmark = pytest.mark.gpu


def test_random_option_sets_match_oracle():
    env = dict(os.environ, GRAFT_REPO_ROOT=util.ROOT)
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tests", "tools", "fuzz_parity.py"), "5", "2000"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "5 cases, 0 mismatches" in r.stdout


def test_random_option_sets_with_random_scheduling_knobs():
    """the same sweep, every case in a context with random band-pool cap / batch size / seeding budget / single-pass threshold:
    lists in chunks, several batches, several seeding chunks - the records must not depend on any of them"""
    env = dict(os.environ, GRAFT_REPO_ROOT=util.ROOT)
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "tests", "tools", "fuzz_parity.py"), "5", "15000", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "5 cases, 0 mismatches" in r.stdout
