#!/usr/bin/env python
"""Generate the golden fixtures of tests/golden/ from the REFERENCE ITSELF.

Run in the build container only (needs oracle/_ref, compiled from /root/reference by oracle/Makefile).
What is committed is data: tiny volume files written by necat_amd.synth (byte-checked against oc2mkdb
here) and the reference's sorted output records for several option sets - no reference source.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from necat_amd import synth  # noqa: E402
from oracle import oracle_api as ora  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

CASES = {
    # name: (dataset, vid, options)
    "a_fast_can_bin": ("vols_a", 0, dict(kmer_size=13, scan_window=20, job=0, binary_output=1)),
    "a_fast_can_txt": ("vols_a", 0, dict(kmer_size=13, scan_window=20, job=0, binary_output=0)),
    "a_fast_m4_txt": ("vols_a", 0, dict(kmer_size=13, scan_window=20, job=1, binary_output=0, use_hdr_as_id=0)),
    "a_fast_m4_bin": ("vols_a", 0, dict(kmer_size=13, scan_window=20, job=1, binary_output=1, use_hdr_as_id=0)),   # 96-byte M4Record
    "a_fast_m4_hdr": ("vols_a", 0, dict(kmer_size=13, scan_window=20, job=1, binary_output=0, use_hdr_as_id=1)),
    "a_sens_m4_txt": ("vols_a", 0, dict(kmer_size=13, scan_window=10, job=1, binary_output=0, use_hdr_as_id=0)),
    "a_k15_m4_txt": ("vols_a", 0, dict(kmer_size=15, scan_window=20, job=1, binary_output=0, use_hdr_as_id=0)),
    "a_topn_m4_txt": ("vols_a", 0, dict(kmer_size=13, scan_window=20, num_candidates=2, job=1, binary_output=0, use_hdr_as_id=0)),
    "b_v0_m4_txt": ("vols_b", 0, dict(kmer_size=12, scan_window=10, kmer_cnt_cutoff=100, job=1, binary_output=0, use_hdr_as_id=0)),
    "b_v1_m4_txt": ("vols_b", 1, dict(kmer_size=12, scan_window=10, kmer_cnt_cutoff=100, job=1, binary_output=0, use_hdr_as_id=0)),
    "b_v2_can_bin": ("vols_b", 2, dict(kmer_size=12, scan_window=10, kmer_cnt_cutoff=100, job=0, binary_output=1)),
}
BASE = dict(kmer_cnt_cutoff=500, block_size=2000, block_score_cutoff=3, num_candidates=500, align_size_cutoff=1000,
            ddfs_cutoff=0.25, error=0.5, num_output=500, num_threads=2, use_hdr_as_id=0)


def write_rel_dir(name, rs, vol_size):
    """volume dir with RELATIVE volume names (tests copy it and rewrite absolute paths)."""
    d = os.path.join(GOLD, name)
    if os.path.exists(d):
        shutil.rmtree(d)
    synth.write_volume_dir(d, rs, vol_size)
    lines = open(os.path.join(d, "volume_names.txt")).read().splitlines()
    with open(os.path.join(d, "volume_names.txt"), "w") as f:
        for ln in lines:
            p, a, b = ln.split()
            f.write("%s\t%s\t%s\n" % (os.path.basename(p), a, b))


def main():
    if not ora.have_ref():
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` in the build container")
    write_rel_dir("vols_a", synth.simulate_reads(40_000, 12.0, seed=101), synth.DEFAULT_VOL_SIZE)
    write_rel_dir("vols_b", synth.simulate_reads(50_000, 14.0, seed=202, err=0.10, repeat_frac=0.1), 260_000)
    tmp = tempfile.mkdtemp(prefix="golden_")
    manifest = {}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    for name, (ds, vid, kw) in CASES.items():
        d = util.install_golden_volumes(ds, tmp)
        o = ora.options(**dict(BASE, **kw))
        out = os.path.join(tmp, name + ".out")
        ora.run_ref(o, vid, d, out)
        recs = ora.sorted_records(out, ora.record_size(o))
        dst = os.path.join(GOLD, name + (".bin" if o.binary_output else ".txt"))
        with open(dst, "wb") as f:
            f.write(b"".join(recs))
        manifest[name] = {"dataset": ds, "vid": vid, "options": dict(BASE, **kw), "records": len(recs),
                          "md5": hashlib.md5(b"".join(recs)).hexdigest(), "file": os.path.basename(dst)}
        print(name, len(recs))
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
