#!/usr/bin/env python
"""Full-size MULTI-VOLUME golden fingerprints from the REFERENCE binary (the shape of BASELINE.json configs[3] / [4]: several
reference volumes, every (reference volume v, query volume i >= v) pair, one volume in the >= 1 Gbp region where 34-bit
offsets, 32-bit slot counts and the candidate batch cap matter).

Run in the build container (needs oracle/_ref/oc2pmov = the reference compiled from /root/reference by oracle/Makefile;
~30 GB of RAM for the 1 Gbp volume's k = 15 table + offset list + sort buffer; about half an hour on 8 cores):

    python tests/golden/make_golden_multivol.py [threads]                 -> tests/golden/multivol_full_reference.json
    python tests/golden/make_golden_multivol.py [threads] drosophila      -> tests/golden/drosophila_full_reference.json
    python tests/golden/make_golden_multivol.py [threads] human_subset    -> tests/golden/human_subset_full_reference.json

`drosophila` is BASELINE.json configs[3] at its real size: a 140 Mb genome x 40 = 5.6 Gbp cut by oc2mkdb's own rule (a volume
is closed once it holds >= 2 000 000 000 bases, makedb/main.c:8,29) into volumes of 2.0 / 2.0 / 1.6 Gbp - six (reference,
query) pairs; ~41 GB of RAM for a 2 Gbp volume's k = 15 table (8.6 GB) + offset list (16 GB) + sort buffer (16 GB), about
two hours on 6 threads.

The dataset is the seeded synthetic one of necat_amd/synth.py (regenerated on the GPU box by the test and fingerprinted by
`reads_md5`), cut into three volumes of UNEQUAL size (synth.write_volume_dir_cuts).  What is committed is data only: per
reference volume the record counts and md5s of the SORTED records of `-j 0 -u 1` (28-byte packed candidates: what the
correction pipeline runs, necat.pl:31-32) and of `-j 1 -u 0 -i 0` (M4 text), and the md5 over all volumes together (= what
oc2pm's concatenated output sorts to).
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

CFGS = {}
CFGS["multivol"] = dict(genome=37_000_000, coverage=40.0, seed=31, err=0.12, cuts=[1_050_000_000, 300_000_000],
           flags="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500")
CFGS["drosophila"] = dict(genome=140_000_000, coverage=40.0, seed=41, err=0.12, cuts=[2_000_000_000, 2_000_000_000],
                          flags="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500")
# BASELINE configs[4] as a stated SUBSET (SURVEY 8d allows one): a 3 Gb genome read at the 30x RATE, of whose 45 oc2mkdb volumes
# the first three are kept - synth.simulate_reads draws reads one after the other until the target is met, so `coverage = 2.0`
# IS the first 6 Gbp of the 90 Gbp read set (volumes 0 and 1 closed by the 2 Gbp rule, volume 2 = what is left of the 6 Gbp).
# The regime no smaller genome has: a 2 Gbp volume is 0.67x of the genome, so its ~2 x 10^9 k-mer positions are nearly all
# distinct (table occupancy ~ 0.85 of 4^15, the -q cut idle) and a read has a handful of true overlaps instead of ~ 80.
CFGS["human_subset"] = dict(genome=3_000_000_000, coverage=2.0, seed=51, err=0.12, cuts=[2_000_000_000, 2_000_000_000],
                            flags="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500")


def main(threads: int, name: str = "multivol") -> None:
    CFG = CFGS[name]
    t0 = time.time()
    rs = synth.simulate_reads(CFG["genome"], CFG["coverage"], seed=CFG["seed"], err=CFG["err"])
    print("%d reads / %d bp generated in %.0f s" % (rs.nreads, rs.nbases, time.time() - t0), flush=True)
    tmp = tempfile.mkdtemp(prefix="necat_mv_", dir=os.environ.get("NECAT_TMP"))
    d = os.path.join(tmp, "vols")
    nvol = synth.write_volume_dir_cuts(d, rs, CFG["cuts"])
    vols = [ln.split("\t") for ln in open(os.path.join(d, "volume_names.txt"))]
    out = {"generator": {k: CFG[k] for k in ("genome", "coverage", "seed", "err", "cuts")}, "options": CFG["flags"],
           "reads_md5": hashlib.md5(rs.codes.tobytes()).hexdigest(), "nreads": rs.nreads, "nbases": rs.nbases, "volumes": nvol,
           "volume_reads": [int(v[2]) for v in vols],
           "source": "oracle/_ref/oc2pmov (the reference compiled from /root/reference), -t %d, per reference volume: -j 0 -u 1 -i 1 and -j 1 -u 0 -i 0" % threads,
           "per_volume": []}
    del rs
    all_can, all_m4 = [], []
    for vid in range(nvol):
        pv = {"volume": vid}
        for mode, extra in (("can", "-j 0 -u 1 -i 1"), ("m4", "-j 1 -u 0 -i 0")):
            res = os.path.join(tmp, "out_%s_%d" % (mode, vid))
            cmd = [ora.REF_PMOV] + CFG["flags"].split() + extra.split() + ["-t", str(threads), d, str(vid), res]
            t0 = time.time()
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            pv[mode + "_reference_wall_s"] = round(time.time() - t0, 1)
            print("  %s: %.0f s" % (" ".join(cmd[1:]), time.time() - t0), flush=True)
            if mode == "m4":
                lines = sorted(open(res, "rb").read().splitlines(keepends=True))
                pv["m4_records"] = len(lines)
                pv["m4_text_sorted_md5"] = hashlib.md5(b"".join(lines)).hexdigest()
                pv["aligned_query_bases"] = sum(int(f[6]) - int(f[5]) for f in (ln.split() for ln in lines))
                all_m4 += lines
            else:
                raw = np.fromfile(res, dtype="<u4").reshape(-1, 7)
                recs = sorted(bytes(r) for r in raw)
                pv["candidate_records"] = len(recs)
                pv["candidates_packed_sorted_md5"] = hashlib.md5(b"".join(recs)).hexdigest()
                all_can += recs
            os.remove(res)
        out["per_volume"].append(pv)
        json.dump(out, open(os.path.join(tmp, "partial.json"), "w"), indent=1)
    all_can.sort(); all_m4.sort()
    out["candidate_records"] = len(all_can)
    out["candidates_packed_sorted_md5"] = hashlib.md5(b"".join(all_can)).hexdigest()
    out["m4_records"] = len(all_m4)
    out["m4_text_sorted_md5"] = hashlib.md5(b"".join(all_m4)).hexdigest()
    shutil.rmtree(tmp, ignore_errors=True)
    path = os.path.join(ROOT, "tests", "golden", "%s_full_reference.json" % name)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1), sys.argv[2] if len(sys.argv) > 2 else "multivol")
