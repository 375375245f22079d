#!/usr/bin/env python
"""Golden vectors for the rescue pair (ocda_go = DALIGNER's local alignment, edlib_go = edlib's NW path), generated from the REFERENCE's own
functions (oracle/_ref/librescue_ref.so = oracle/rescue_ref_shim.c over the reference objects).  Committed is data only:
tests/golden/rescue_cases.json - per case the seed the test rebuilds the read pair from (tests/test_rescue.py::read_pair) and what the reference
returned: end points, difference count, identity and, for edlib_go, 64-bit FNV hashes of the two alignment strings.

    python tests/golden/make_golden_rescue.py
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import test_rescue as T  # noqa: E402
from oracle import oracle_api as ora  # noqa: E402


def main():
    if not os.path.exists(T.REF):
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` in the build container")
    ref = C.CDLL(T.REF)
    out = {"ocda_go": [], "edlib_go": []}
    for seed in range(2000, 2060):
        c = T.ocda_inputs(seed)
        if c is None:
            continue
        q, t, qs, ts, e = c
        o, ident = (C.c_int * 6)(), C.c_double()
        r = ref.ref_ocda_go(T.ptr(q), qs, len(q), T.ptr(t), ts, len(t), C.c_double(e), 100, o, C.byref(ident))
        out["ocda_go"].append({"seed": seed, "ret": r, "out": list(o)[:5], "ident": ident.value if r else 0.0})
    for seed in range(3000, 3060):
        c = T.edlib_inputs(seed, big=seed % 4 == 3)
        if c is None:
            continue
        q, t, qf, qt, tf, tt, error, tol = c
        cap = (qt - qf) + (tt - tf) + 16
        o, ident = (C.c_int * 6)(), C.c_double()
        qa, ta = C.create_string_buffer(cap), C.create_string_buffer(cap)
        r = ref.ref_edlib_go(T.ptr(q), qf, qt, T.ptr(t), tf, tt, C.c_double(error), tol, 100, o, C.byref(ident), qa, ta, cap)
        rec = {"seed": seed, "ret": r}
        if r:
            rec.update({"out": list(o), "ident": ident.value, "qaln": ora.fnv64(qa.value), "taln": ora.fnv64(ta.value)})
        out["edlib_go"].append(rec)
    with open(os.path.join(ROOT, "tests", "golden", "rescue_cases.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(len(out["ocda_go"]), "ocda_go cases,", sum(c["ret"] for c in out["ocda_go"]), "aligned;", len(out["edlib_go"]), "edlib_go cases,",
          sum(1 for c in out["edlib_go"] if c["ret"]), "aligned")


if __name__ == "__main__":
    main()
