#!/usr/bin/env python
"""Golden vectors of onc_align WITH its gapped strings, produced by the REFERENCE ITSELF
(oracle/_ref/libnecat_ref.so = /root/reference's own sources, built by oracle/Makefile).

Inputs: the candidates of tests/golden/a_fast_can_txt.txt on tests/golden/vols_a (both data files are
fixtures already).  Output: tests/golden/onc_align_a.json - per candidate and tail_match_len (4 = what
oc2cns passes, 1 = what oc2pmov passes): return value, coordinates, identity, md5 of both gapped strings.

    python tests/golden/make_golden_onc_align.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from necat_amd import synth  # noqa: E402
from oracle import oracle_api as ora  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
N_CAND = 160


def volume_codes(path):
    pac, offs, sizes, _ = synth.read_volume(path)
    nb = int(offs[-1] + sizes[-1])
    codes = (np.repeat(pac, 4)[:nb] >> ((3 - (np.arange(nb) & 3)) * 2)) & 3
    return codes.astype(np.uint8), offs, sizes


def cases():
    codes, offs, sizes = volume_codes(os.path.join(GOLD, "vols_a", "vol0"))
    cand = np.loadtxt(os.path.join(GOLD, "a_fast_can_txt.txt"), dtype=np.int64)[:N_CAND]
    for r in cand:   # columns: qid sid score qdir qbeg qend qoff qsize sdir sbeg send soff ssize
        qid, sid, qdir, qoff, soff = int(r[0]), int(r[1]), int(r[3]), int(r[6]), int(r[11])
        q = codes[offs[qid]:offs[qid] + sizes[qid]]
        if qdir == 1:
            q = (3 - q[::-1]).astype(np.uint8)
        t = codes[offs[sid]:offs[sid] + sizes[sid]]
        yield (qid, sid, qdir, qoff, soff), q, t


def run(impl):
    al = ora.Aligner(0.5, impl)
    out = []
    for tail in (4, 1):
        for key, q, t in cases():
            ok, qoff, qend, toff, tend, ident, qa, ta = al.align(q, key[3], t, key[4], 1000, tail)
            out.append(dict(cand=list(key), tail=tail, ok=int(ok), qoff=qoff, qend=qend, toff=toff, tend=tend,
                            ident="%.17g" % ident, columns=len(qa),
                            query_align_md5=hashlib.md5(qa).hexdigest(), target_align_md5=hashlib.md5(ta).hexdigest()))
    al.close()
    return out


if __name__ == "__main__":
    if not os.path.exists(ora.REF_LIB):
        raise SystemExit("oracle/_ref/libnecat_ref.so missing: run `make -C oracle ref` in the build container")
    with open(os.path.join(GOLD, "onc_align_a.json"), "w") as f:
        json.dump(run("ref"), f, indent=0)
    print("wrote onc_align_a.json")
