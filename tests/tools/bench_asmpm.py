#!/usr/bin/env python
"""(Under tests/: it runs the reference's program from oracle/_ref as the comparison.)  oc2asmpm on a synthetic corrected-read set: this repo's program (GPU block aligner) and, when oracle/_ref is there, the reference's own program on all
host cores; checks that the records are the same and prints the wall times.  NECAT_TRACE=2 shows the library's stage times.

    python tests/tools/bench_asmpm.py [genome_len] [coverage] [error]
"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from necat_amd import build, synth  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    cov = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
    err = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
    build.build_cli()
    tmp = tempfile.mkdtemp(prefix="asmpm_bench_")
    rs = synth.simulate_reads(G, cov, seed=71, err=err, repeat_frac=0.05)
    wrk = os.path.join(tmp, "vols")
    nv = synth.write_volume_dir(wrk, rs)
    args = "-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400".split()
    cores = os.cpu_count() or 1
    t = time.time()
    r = subprocess.run([build.OC2ASMPM] + args + ["-t", str(min(cores, 32)), wrk, "0", os.path.join(tmp, "mine.m4")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    t_mine = time.time() - t
    print(r.stdout[-1500:])
    assert r.returncode == 0
    n = len(open(os.path.join(tmp, "mine.m4")).read().splitlines())
    print("reads %d (%d bases, %d volume(s)), records %d, this repo %.2f s" % (rs.nreads, rs.nbases, nv, n, t_mine))
    ref = os.path.join(ROOT, "oracle", "_ref", "oc2asmpm")
    if os.path.exists(ref):
        t = time.time()
        subprocess.run([ref] + args + ["-t", str(cores), wrk, "0", os.path.join(tmp, "ref.m4")], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        t_ref = time.time() - t
        same = sorted(open(os.path.join(tmp, "ref.m4")).read().splitlines()) == sorted(open(os.path.join(tmp, "mine.m4")).read().splitlines())
        print("reference on %d threads %.2f s; same records: %s" % (cores, t_ref, same))


if __name__ == "__main__":
    main()
