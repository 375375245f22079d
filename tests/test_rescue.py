"""The rescue pair (necat_amd/csrc/rescue.h: ocda_go = DALIGNER's local alignment, edlib_go = edlib's NW path) against the
REFERENCE's own functions, linked from /root/reference into oracle/_ref/librescue_ref.so (oracle/rescue_ref_shim.c): end points,
difference counts, identities and - for edlib_go - the whole trimmed alignment strings must be identical on random read pairs
with substitutions, indels, long gaps and unrelated flanks."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import util

REF = os.path.join(util.ROOT, "oracle", "_ref", "librescue_ref.so")
needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/librescue_ref.so (the reference's build) is absent")


@pytest.fixture(scope="module")
def mine_lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("rescue")
    so = os.path.join(str(d), "librescue_mine.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so,
                    os.path.join(util.ROOT, "tests", "host_core", "rescue_capi.cpp")], check=True)
    return C.CDLL(so)


@pytest.fixture(scope="module")
def libs(mine_lib):
    return C.CDLL(REF), mine_lib


def mutate(t, rng, err, gap=None):
    q = []
    for i, c in enumerate(t):
        if gap and gap[0] <= i < gap[0] + gap[1]:
            continue
        r = rng.random()
        if r < err / 3:
            q.append((c + 1 + rng.integers(0, 3)) % 4)
        elif r < 2 * err / 3:
            q.append(c)
            q.append(rng.integers(0, 4))
        elif r < err:
            continue
        else:
            q.append(c)
    return np.array(q, dtype=np.uint8)


def read_pair(seed, lo_len=800, hi_len=9000):
    rng = np.random.default_rng(seed)
    L = int(rng.integers(lo_len, hi_len))
    t = rng.integers(0, 4, L, dtype=np.uint8)
    err = float(rng.choice([0.05, 0.15, 0.25, 0.35]))
    gap = (int(rng.integers(100, L - 100)), int(rng.integers(50, 600))) if rng.random() < 0.5 else None
    lo = int(rng.integers(0, L // 3))
    hi = int(rng.integers(2 * L // 3, L))
    q = mutate(t[lo:hi], rng, err, (gap[0] - lo, gap[1]) if gap and lo < gap[0] < hi else None)
    if rng.random() < 0.3:
        q = np.concatenate([rng.integers(0, 4, int(rng.integers(0, 400)), dtype=np.uint8), q,
                            rng.integers(0, 4, int(rng.integers(0, 400)), dtype=np.uint8)])
    return rng, q, t, lo


def ptr(a):
    return a.ctypes.data_as(C.c_char_p)


def ocda_inputs(seed):
    rng, q, t, lo = read_pair(seed)
    if len(q) < 300:
        return None
    qs = int(rng.integers(50, len(q) - 50))
    ts = min(len(t) - 1, max(0, lo + qs + int(rng.integers(-30, 30))))
    return q, t, qs, ts, float(rng.choice([0.5, 0.3, 0.2]))


def edlib_inputs(seed, big):
    rng, q, t, lo = read_pair(seed, 300 if not big else 3000, 3000 if not big else 9000)
    if len(q) < 200:
        return None
    qf = int(rng.integers(0, 40))
    qt = len(q) - int(rng.integers(0, 40))
    tf = max(0, lo + int(rng.integers(-20, 20)))
    tt = min(len(t), lo + int((qt - qf) * rng.uniform(0.9, 1.1)) + int(rng.integers(0, 300)))
    if tt - tf < 150:
        return None
    return q, t, qf, qt, tf, tt, float(rng.choice([0.5, 0.3])), int(rng.choice([0.1, 0.3, 0.6]) * (qt - qf)) + 10


def test_rescue_golden(mine_lib):
    """the committed vectors (tests/golden/rescue_cases.json: what the reference's ocda_go / edlib_go returned; make_golden_rescue.py)"""
    import json
    from oracle import oracle_api as ora
    g = json.load(open(os.path.join(util.GOLDEN, "rescue_cases.json")))
    n_ok = 0
    for c in g["ocda_go"]:
        q, t, qs, ts, e = ocda_inputs(c["seed"])
        o, ident = (C.c_int * 6)(), C.c_double()
        r = mine_lib.mine_ocda_go(ptr(q), qs, len(q), ptr(t), ts, len(t), C.c_double(e), 100, o, C.byref(ident))
        assert (r, list(o)[:5]) == (c["ret"], c["out"]), c["seed"]
        if r:
            assert ident.value == c["ident"], c["seed"]
            n_ok += 1
    assert n_ok > 20
    n_ok = 0
    for c in g["edlib_go"]:
        q, t, qf, qt, tf, tt, error, tol = edlib_inputs(c["seed"], c["seed"] % 4 == 3)
        cap = (qt - qf) + (tt - tf) + 16
        o, ident = (C.c_int * 6)(), C.c_double()
        qa, ta = C.create_string_buffer(cap), C.create_string_buffer(cap)
        r = mine_lib.mine_edlib_go(ptr(q), qf, qt, ptr(t), tf, tt, C.c_double(error), tol, 100, o, C.byref(ident), qa, ta, cap)
        assert r == c["ret"], c["seed"]
        if r:
            assert (list(o), ident.value, ora.fnv64(qa.value), ora.fnv64(ta.value)) == (c["out"], c["ident"], c["qaln"], c["taln"]), c["seed"]
            n_ok += 1
    assert n_ok > 15


@needs_ref
def test_ocda_go_matches_reference(libs):
    ref, mine = libs
    n = ok = 0
    for seed in range(160):
        rng, q, t, lo = read_pair(seed)
        if len(q) < 300:
            continue
        qs = int(rng.integers(50, len(q) - 50))
        ts = min(len(t) - 1, max(0, lo + qs + int(rng.integers(-30, 30))))
        e = float(rng.choice([0.5, 0.3, 0.2]))
        o1, o2, i1, i2 = (C.c_int * 6)(), (C.c_int * 6)(), C.c_double(), C.c_double()
        a = [ptr(q), qs, len(q), ptr(t), ts, len(t), C.c_double(e), 100]
        r1 = ref.ref_ocda_go(*a, o1, C.byref(i1))
        r2 = mine.mine_ocda_go(*a, o2, C.byref(i2))
        assert (r1, list(o1)[:5]) == (r2, list(o2)[:5]), seed
        if r1:
            assert i1.value == i2.value, seed
            ok += 1
        n += 1
    assert n > 120 and ok > 60


def edlib_case(ref, mine, q, t, qf, qt, tf, tt, error, tol, min_size=100):
    cap = (qt - qf) + (tt - tf) + 16
    res = []
    for f in (ref.ref_edlib_go, mine.mine_edlib_go):
        o, ident = (C.c_int * 6)(), C.c_double()
        qa, ta = C.create_string_buffer(cap), C.create_string_buffer(cap)
        r = f(ptr(q), qf, qt, ptr(t), tf, tt, C.c_double(error), tol, min_size, o, C.byref(ident), qa, ta, cap)
        res.append((r,) if not r else (r, list(o), ident.value, qa.value, ta.value))
    return res


@needs_ref
def test_edlib_go_matches_reference(libs):
    """Sizes on both sides of edlib's 1 MB traceback limit, so that the plain walk, one split and nested splits all occur."""
    ref, mine = libs
    n = ok = 0
    for seed in range(140):
        big = seed % 4 == 3
        rng, q, t, lo = read_pair(1000 + seed, 300 if not big else 3000, 3000 if not big else 9000)
        if len(q) < 200:
            continue
        # the range a local alignment would hand over, plus cases that start / end off the alignment
        qf = int(rng.integers(0, 40))
        qt = len(q) - int(rng.integers(0, 40))
        tf = max(0, lo + int(rng.integers(-20, 20)))
        tt = min(len(t), lo + int((qt - qf) * rng.uniform(0.9, 1.1)) + int(rng.integers(0, 300)))
        if tt - tf < 150:
            continue
        error = float(rng.choice([0.5, 0.3]))
        tol = int(rng.choice([0.1, 0.3, 0.6]) * (qt - qf)) + 10
        a, b = edlib_case(ref, mine, q, t, qf, qt, tf, tt, error, tol)
        assert a == b, seed
        n += 1
        ok += a[0]
    assert n > 100 and ok > 40


@needs_ref
def test_edlib_go_edges(libs):
    ref, mine = libs
    rng = np.random.default_rng(5)
    t = rng.integers(0, 4, 5000, dtype=np.uint8)
    q = t.copy()
    cases = [
        (q, t, 0, 5000, 0, 5000, 0.5, 100),      # identical
        (q, t, 0, 5000, 0, 5000, 0.5, 0),        # tolerance 0 with distance 0
        (q, t, 0, 4000, 0, 5000, 0.5, 999),      # length difference above the tolerance
        (q, t, 0, 4000, 0, 5000, 0.5, 1000),     # all of the difference is one end gap
        (q, t, 100, 4100, 0, 5000, 0.5, 2000),   # gaps at both ends
        (q, t, 0, 120, 0, 90, 0.5, 60),          # shorter than min_align_size
        (q, t, 0, 64, 0, 4800, 1e9, 5000),       # one block of rows, thousands of columns
        (q, t, 0, 4800, 0, 130, 1e9, 5000),      # and the transpose
    ]
    q2 = mutate(t, rng, 0.4)
    cases.append((q2, t, 0, len(q2), 0, 5000, 0.2, 4000))    # distance within the tolerance but above error * length
    q3 = rng.integers(0, 4, 5000, dtype=np.uint8)
    cases.append((q3, t, 0, 5000, 0, 5000, 1e9, 5000))       # unrelated sequences: runs of 4 matches still end the trimming
    for i, c in enumerate(cases):
        a, b = edlib_case(ref, mine, *c)
        assert a == b, i


@needs_ref
def test_cns_loop_with_rescue_matches_reference(tmp_path):
    """oc2cns -r 1 without a GPU: the extension loop (cns_loop.h) with the oracle's block-wise aligner and cns_rescue.h behind it,
    as the library runs it behind the device pass, on reads with long indels - every add_one_align call (gapped strings included)
    and every template's numbers as logged from the REFERENCE's own consensus driver run with -r 1."""
    from oracle import oracle_api as ora
    if not ora.have_ref_cns():
        pytest.skip("oracle/_ref/cns_ref_harness is absent")
    d = str(tmp_path)
    objs = []
    for src in ("necat_oracle.c", "cns_oracle.c"):
        obj = os.path.join(d, src[:-2] + ".o")
        subprocess.run(["gcc", "-O2", "-std=gnu99", "-c", os.path.join(util.ROOT, "oracle", src), "-o", obj], check=True)
        objs.append(obj)
    exe = os.path.join(d, "check_cns")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                    os.path.join(util.ROOT, "tests", "host_core", "check_cns.cpp")] + objs + ["-lm", "-lpthread"], check=True)
    wrk, can, _ = util.make_long_indel_partition(tmp_path)
    want = os.path.join(d, "ref_r1.txt")
    subprocess.run([ora.REF_CNS] + ora.cns_argv(ora.cns_options()) + ["-r", "1", wrk, can, want, "full"], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    got = os.path.join(d, "mine_r1.txt")
    r = subprocess.run([exe, wrk, can] + "400 4 12 0.5 0.8 0 1 12 1.25 2 1".split() + [got], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    f = dict(kv.split("=") for kv in r.stdout.split())
    assert int(f["rescued"]) > 150 and int(f["templates"]) > 50
    assert open(got).read() == open(want).read()
