#!/usr/bin/env python
"""Golden vectors for oc2asmpm and oc2rm_worker (8f.4), generated from the REFERENCE ITSELF.

Run in the build container only (needs oracle/_ref: `make -C oracle ref`).  Committed is data only:
  tests/golden/vols_d/           corrected-read-like volumes (necat_amd.synth; 30 kb genome x 10, 3 % error, a third of the reads with long indels)
  tests/golden/asm_d/ref_v<i>.m4 what the reference's `oc2asmpm -n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400 -k 13 -t 1 vols_d <i> out` wrote
  tests/golden/vols_e/ + rm_e/ref.vol    raw reads with long indels + the reference volume they are mapped to (three contigs + one unrelated sequence)
  tests/golden/rm_e/ref.m4       what the reference's `oc2rm_worker -k 13 -i 0 -t 1 vols_e ref.vol out` wrote

    python tests/golden/make_golden_asm_rm.py
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from necat_amd import synth  # noqa: E402
from oracle import oracle_api as ora  # noqa: E402
import util  # noqa: E402
from make_golden import write_rel_dir  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
ASM_ARGS = "-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400 -k 13"
RM_ARGS = "-k 13 -i 0"


def main():
    ref_asm = os.path.join(os.path.dirname(ora.REF_PMOV), "oc2asmpm")
    if not (os.path.exists(ref_asm) and os.path.exists(ora.REF_RM)):
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` in the build container")
    tmp = tempfile.mkdtemp(prefix="golden_asm_rm_")
    manifest = {}
    # oc2asmpm
    rs = synth.add_long_indels(synth.simulate_reads(30_000, 10.0, seed=404, err=0.03, repeat_frac=0.3), 0.3, seed=405)
    write_rel_dir("vols_d", rs, 150_000)
    wrk = util.install_golden_volumes("vols_d", tmp)
    nv = len(open(os.path.join(wrk, "volume_names.txt")).read().splitlines())
    dst = os.path.join(GOLD, "asm_d")
    os.makedirs(dst, exist_ok=True)
    n = 0
    for v in range(nv):
        out = os.path.join(dst, "ref_v%d.m4" % v)
        subprocess.run([ref_asm] + ASM_ARGS.split() + ["-t", "1", wrk, str(v), out], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        n += len(open(out).read().splitlines())
    manifest["asm_d"] = {"volumes": "vols_d", "n_volumes": nv, "args": ASM_ARGS, "records": n}
    # oc2rm_worker
    seed, glen = 505, 60_000
    G = synth.make_genome(glen, seed, 0.4)
    rs = synth.add_long_indels(synth.simulate_reads(coverage=6.0, seed=seed, err=0.12, genome=G), 0.5, seed=seed + 1, lo=300, hi=1500)
    write_rel_dir("vols_e", rs, 150_000)
    wrk = util.install_golden_volumes("vols_e", tmp)
    dst = os.path.join(GOLD, "rm_e")
    os.makedirs(dst, exist_ok=True)
    cuts = [0, glen * 4 // 15, glen * 11 // 15, glen]
    seqs = [G[cuts[i]:cuts[i + 1]] for i in range(3)] + [np.random.default_rng(seed + 2).integers(0, 4, 8_000, dtype=np.uint8)]
    ref = os.path.join(dst, "ref.vol")
    synth.write_volume(ref, np.concatenate(seqs), [len(x) for x in seqs], ["ctg%d" % i for i in range(4)])
    out = os.path.join(dst, "ref.m4")
    subprocess.run([ora.REF_RM] + RM_ARGS.split() + ["-t", "1", wrk, ref, out], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    manifest["rm_e"] = {"volumes": "vols_e", "args": RM_ARGS, "records": len(open(out).read().splitlines())}
    with open(os.path.join(GOLD, "manifest_asm_rm.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(manifest)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
