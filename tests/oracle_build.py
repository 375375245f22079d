"""Build the test oracle (oracle/Makefile): liboracle.so, oc2pmov_oracle and - when /root/reference is present - the
reference itself under oracle/_ref.  TEST INFRASTRUCTURE: used by tests/, __graft_entry__ and bench.py's cpu_baseline
leg only; building the checker is not using it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")


def _make(*args):
    r = subprocess.run(["make", "-s", "-C", ORACLE_DIR] + list(args), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("oracle build failed: make %s" % " ".join(args))


def build_oracle(force: bool = False) -> str:
    if force:
        _make("clean")
    _make("all")
    return ORACLE_LIB


if __name__ == "__main__":
    print("built:", build_oracle("--force" in sys.argv))
