#!/usr/bin/env python
"""Full-size golden fingerprints (BASELINE.json configs[1] and configs[2]) from the REFERENCE binary.

Run in the build container (needs oracle/_ref/oc2pmov = the reference compiled from /root/reference by
oracle/Makefile; ~10 GB of RAM per run for the k = 15 table):

    python tests/golden/make_golden_full.py ecoli      -> tests/golden/ecoli_full_reference.json
    python tests/golden/make_golden_full.py yeast      -> tests/golden/yeast_full_reference.json
    python tests/golden/make_golden_full.py ecoli_repeats | ecoli_err6 | ecoli_longtail   (round 6: SURVEY 8d's other data shapes)

The dataset is the seeded synthetic one of necat_amd/synth.py (regenerated on the GPU box by the tests and
fingerprinted by `reads_md5`); what is committed is data only: record counts and md5s of the SORTED records
(28-byte packed candidates of `-j 0 -u 1`; 12-column M4 text lines of `-j 1 -u 0 -i 0`), because the
reference's threads flush records in no particular order (record_writer.c:40-46).
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from necat_amd import synth                      # noqa: E402
from oracle import oracle_api as ora             # noqa: E402

CONFIGS = {
    # BASELINE configs[1]: E. coli 4.6 Mb x 40, OVLP_FAST_OPTIONS
    "ecoli": dict(genome=4_600_000, coverage=40.0, seed=7, err=0.12,
                  flags="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"),
    # BASELINE configs[2]: S. cerevisiae-size 12 Mb x 50, OVLP_SENSITIVE_OPTIONS (-z 10)
    "yeast": dict(genome=12_000_000, coverage=50.0, seed=13, err=0.12,
                  flags="-k 15 -z 10 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"),
    # SURVEY 8d's other data shapes at configs[1]'s size (round 6).  Repeat families (unit length, copies): 5 kb x 140 - each of its 15-mers
    # occurs ~ 40 x 140 x 0.88^15 = 820 times in the volume, above -q 500: dropped by the cutoff (lookup_table.c:15-58) -, 20 kb x 24 and
    # 2 kb x 60 (~ 140 / 350 occurrences: kept; blocks at their 40 seeds, word_finder.c:91-92, reads with far more than -n 500 candidates,
    # pm_worker.c:139-140, :168-171): 29 % of the genome inside a repeat
    "ecoli_repeats": dict(genome=4_600_000, coverage=40.0, seed=31, err=0.12, families=[[5000, 140], [20000, 24], [2000, 60]],
                          flags="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"),
    # 6 % errors: twice the exact 15-mers per overlap, candidate counts and chain lengths change by > 2 x, blocks of distance ~ 55
    "ecoli_err6": dict(genome=4_600_000, coverage=40.0, seed=37, err=0.06,
                       flags="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"),
    # long-tailed read lengths: log-normal (median 6 kb, sigma 0.85), cut at 200 kb - a few reads of 100 - 180 kb, chains of 300 blocks
    "ecoli_longtail": dict(genome=4_600_000, coverage=40.0, seed=29, err=0.12, lognormal=[6000, 0.85], max_len=200_000,
                           flags="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"),
}
GEN_KEYS = ("genome", "coverage", "seed", "err", "families", "lognormal", "max_len")


def generate(g: dict):
    """the dataset of a golden's "generator" entry (the tests call this too)"""
    return synth.simulate_reads(g["genome"], g["coverage"], seed=g["seed"], err=g["err"], families=g.get("families"),
                                lognormal=g.get("lognormal"), max_len=g.get("max_len"))


def run(name: str, threads: int) -> None:
    cfg = CONFIGS[name]
    t0 = time.time()
    rs = generate(cfg)
    print("%s: %d reads / %d bp generated in %.0f s" % (name, rs.nreads, rs.nbases, time.time() - t0), flush=True)
    tmp = tempfile.mkdtemp(prefix="necat_full_")
    d = os.path.join(tmp, "vols")
    nvol = synth.write_volume_dir(d, rs)
    assert nvol == 1
    out = {"generator": {k: cfg[k] for k in GEN_KEYS if k in cfg}, "options": cfg["flags"],
           "reads_md5": hashlib.md5(rs.codes.tobytes()).hexdigest(), "nreads": rs.nreads, "nbases": rs.nbases,
           "source": "oracle/_ref/oc2pmov (the reference compiled from /root/reference), -t %d, -j 1 -u 0 -i 0 and -j 0 -u 1" % threads}
    for mode, extra in (("m4", "-j 1 -u 0 -i 0"), ("can", "-j 0 -u 1 -i 1")):
        res = os.path.join(tmp, "out_" + mode)
        cmd = [ora.REF_PMOV] + cfg["flags"].split() + extra.split() + ["-t", str(threads), d, "0", res]
        t0 = time.time()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        print("  %s: %.0f s" % (" ".join(cmd[1:]), time.time() - t0), flush=True)
        if mode == "m4":
            lines = sorted(open(res, "rb").read().splitlines(keepends=True))
            out["m4_records"] = len(lines)
            out["m4_text_sorted_md5"] = hashlib.md5(b"".join(lines)).hexdigest()
            qb = 0
            for ln in lines:                       # columns: qid sid ident vscore qdir qoff qend qsize sdir soff send ssize
                f = ln.split()
                qb += int(f[6]) - int(f[5])
            out["aligned_query_bases"] = qb
        else:
            raw = np.fromfile(res, dtype="<u4").reshape(-1, 7)
            recs = sorted(bytes(r) for r in raw)
            out["candidate_records"] = len(recs)
            out["candidates_packed_sorted_md5"] = hashlib.md5(b"".join(recs)).hexdigest()
        os.remove(res)
    shutil.rmtree(tmp, ignore_errors=True)
    path = os.path.join(ROOT, "tests", "golden", "%s_full_reference.json" % name)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "ecoli"
    run(which, int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1))
