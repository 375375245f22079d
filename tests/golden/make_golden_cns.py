#!/usr/bin/env python
"""Golden vectors for the consensus stage's extension loop, generated from the REFERENCE ITSELF.

Run in the build container only (needs oracle/_ref: `make -C oracle ref`).  Committed is data only:
  tests/golden/vols_c/            tiny volume files (necat_amd.synth; 20 kb genome x 40)
  tests/golden/cns_c/cands.p0     candidate partition = reference oc2pmov -j 0 -u 1 piped through reference oc2pcan
  tests/golden/cns_c/cands.partitions
  tests/golden/cns_c/ref_<case>.txt  what the reference's consensus_one_partition decided per template, logged by
                                  oracle/cns_ref_harness.c (every add_one_align call + CnsSeq numbers + cov_ranges)

    python tests/golden/make_golden_cns.py
"""
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from necat_amd import synth  # noqa: E402
from oracle import oracle_api as ora  # noqa: E402
import util  # noqa: E402
from make_golden import write_rel_dir  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = {"default": {}, "fixed": dict(use_fixed_ident_cutoff=1), "cov6": dict(max_cov=6, min_cov=2),
         "a2000": dict(min_align_size=2000, mapping_ratio=0.5)}
# the reference's oc2cns itself on the partition: (CnsOptions overrides, extra argv)
OC2CNS_CASES = {"default": ({}, []), "full": ({}, ["-f", "1"]), "cov6_l300": (dict(max_cov=6, min_cov=2), ["-l", "300"]),
                "x9_l3000": (dict(min_cov=9, max_cov=10), ["-l", "3000"])}
FULL = ()      # the gapped strings are logged as 64-bit FNV hashes (a full log is 32 MB)


def main():
    if not ora.have_ref_cns():
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` in the build container")
    write_rel_dir("vols_c", synth.simulate_reads(20_000, 40.0, seed=303, err=0.12), 500_000)
    tmp = tempfile.mkdtemp(prefix="golden_cns_")
    wrk = util.install_golden_volumes("vols_c", tmp)
    nv = len(open(os.path.join(wrk, "volume_names.txt")).read().splitlines())
    o = ora.options(**dict(util.FAST, job=0, binary_output=1, num_threads=1))
    can = os.path.join(tmp, "cands")
    with open(can, "wb") as f:
        for v in range(nv):
            ora.run_ref(o, v, wrk, can + ".v%d" % v)
            f.write(open(can + ".v%d" % v, "rb").read())
    ora.run_ref_pcan(wrk, can)
    dst = os.path.join(GOLD, "cns_c")
    os.makedirs(dst, exist_ok=True)
    for fn in ("cands.p0", "cands.partitions"):
        shutil.copy(os.path.join(tmp, fn), os.path.join(dst, fn))
    manifest = {"volumes": "vols_c", "n_volumes": nv, "candidates": os.path.getsize(can) // 28, "cases": {}}
    for name, kw in CASES.items():
        log = os.path.join(dst, "ref_%s.txt" % name)
        ora.run_ref_cns(ora.cns_options(**kw), wrk, can, log, full=name in FULL)
        L = ora.parse_cns_log(log)
        manifest["cases"][name] = {"options": kw, "templates": len(L), "overlaps": sum(len(t[6]) for t in L), "full": name in FULL}
        print(name, manifest["cases"][name])
    # the whole stage: the reference's oc2cns on the same partition (-t 1: records in template order); committed are the
    # md5s and sizes of its two output files
    import hashlib
    import subprocess
    manifest["oc2cns"] = {}
    for name, (kw, extra) in OC2CNS_CASES.items():
        o = ora.cns_options(**kw)
        co, ro = os.path.join(tmp, "cns_" + name), os.path.join(tmp, "raw_" + name)
        subprocess.run([ora.REF_OC2CNS] + ora.cns_argv(o) + extra + ["-t", "1", wrk, can, co, ro], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        a, b = open(co, "rb").read(), open(ro, "rb").read()
        manifest["oc2cns"][name] = {"options": kw, "extra_argv": extra, "cns_md5": hashlib.md5(a).hexdigest(), "cns_bytes": len(a), "cns_records": a.count(b">"),
                                    "raw_md5": hashlib.md5(b).hexdigest(), "raw_bytes": len(b), "raw_records": b.count(b">")}
        print("oc2cns", name, manifest["oc2cns"][name])
    with open(os.path.join(dst, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
