"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/necat_hip.h declares; without a GPU every entry point fails loudly (no CPU fallback); the
oc2pmov / oc2pm programs keep the reference's argv contract."""
import ctypes
import os
import re
import subprocess

import pytest

from tests import util

try:
    import torch
    HAVE_GPU = torch.cuda.is_available()
except Exception:      # pragma: no cover
    HAVE_GPU = False


def test_library_exports_every_declared_symbol(built):
    from necat_amd import capi
    hdr = open(os.path.join(util.ROOT, "include", "necat_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(necat_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 14
    for path in (built.LIB, built.build_xcheck()):          # the product and the tests' cross-check build of the same sources: one ABI
        lib = ctypes.CDLL(path)
        for name in declared:
            assert hasattr(lib, name), (path, name)
    assert sorted(capi.EXPORTED_SYMBOLS) == declared


def test_the_product_library_has_no_retired_kernels(built):
    """the kernel families the default paths replaced (k_myers_coop, k_myers, k_myers_a16, the band-record k_traceback forms, k_walk_wave, k_rcwalk2 / 4,
    the lane-per-strand seed collection, k_asm_align) exist in libnecat_hip_xcheck.so only; which library a knob set needs: capi.needs_xcheck"""
    import subprocess
    from necat_amd import capi

    def stubs(path):
        out = subprocess.run(["nm", "-C", path], stdout=subprocess.PIPE, text=True).stdout
        return set(re.findall(r"__device_stub__(k_[a-z0-9_]+)", out))
    prod, xc = stubs(built.LIB), stubs(built.build_xcheck())
    retired = {"k_myers_coop", "k_myers", "k_myers_a16", "k_walk_wave", "k_rcwalk2", "k_rcwalk4", "k_seed_collect", "k_asm_align", "k_items_hist"}
    assert retired <= xc and not (retired & prod) and prod < xc
    assert {"k_myers_ck", "k_rcwalk3", "k_rcwalk2w", "k_tail_fused", "k_seed_collect_wave", "k_slice_emit"} <= prod
    # the rule the test binding uses to pick the library
    assert not capi.needs_xcheck({}) and not capi.needs_xcheck({"NECAT_RCWALK": "1", "NECAT_TAIL_FUSED": "0"}) and not capi.needs_xcheck({"NECAT_TAIL_FUSED": "100000000"})
    assert not capi.needs_xcheck({"NECAT_RC_POOL_MB": "1", "NECAT_CK_POST": "0", "NECAT_RC_FASTB": "0", "NECAT_RC_MERGE": "0", "NECAT_INDEX_LDS": "0", "NECAT_CHAIN_WAVE": "0"})
    for env in ({"NECAT_RCWALK": "0"}, {"NECAT_TAIL_FUSED": "0"}, {"NECAT_RC_CARRY": "0"}, {"NECAT_RC_RAGGED": "0"}, {"NECAT_RC_LISTB": "0"}, {"NECAT_RC_WW": "0"},
                {"NECAT_FAST": "0"}, {"NECAT_COOP_THRESHOLD": "0"}, {"NECAT_RC_MAXDIST": "90"}, {"NECAT_SEED_WAVE": "0"}, {"NECAT_ASM_LANE": "1"}, {"NECAT_ASM_RC": "0"}):
        assert capi.needs_xcheck(env), env


def test_struct_layouts_match_reference_records(built):
    from necat_amd import capi
    assert capi.M4_DTYPE.itemsize == 96          # m4_record.h:10-25
    assert capi.CANDIDATE_DTYPE.itemsize == 88   # gapped_candidate.h:9-19
    o = capi.default_options()
    assert (o.kmer_size, o.scan_window, o.kmer_cnt_cutoff, o.block_size, o.block_score_cutoff) == (15, 10, 500, 2000, 3)
    assert (o.num_candidates, o.align_size_cutoff, o.job, o.binary_output, o.use_hdr_as_id) == (500, 500, 1, 0, 1)


@pytest.mark.skipif(HAVE_GPU, reason="a GPU is present")
def test_no_silent_cpu_fallback(built):
    from necat_amd import capi
    with pytest.raises(capi.NecatError):
        capi.Context(0)


def test_pack_candidates_matches_oracle_layout(built):
    """host-side record packing (gapped_candidate.c:13-30) against the golden binary records."""
    import json
    import numpy as np
    from necat_amd import capi
    man = json.load(open(os.path.join(util.GOLDEN, "manifest.json")))
    txt = open(os.path.join(util.GOLDEN, man["a_fast_can_txt"]["file"])).read().split("\n")
    rows = [tuple(int(x) for x in ln.split()) for ln in txt if ln]
    c = np.zeros(len(rows), dtype=capi.CANDIDATE_DTYPE)
    for i, r in enumerate(rows):   # qid sid score qdir qbeg qend qoff qsize sdir sbeg send soff ssize
        c[i] = (r[0], r[1], r[3], r[8], r[2], 0, r[4], r[5], r[7], r[9], r[10], r[12], r[6], r[11])
    packed = sorted(bytes(x) for x in capi.pack_candidates(c).astype("<u4"))
    gold = open(os.path.join(util.GOLDEN, man["a_fast_can_bin"]["file"]), "rb").read()
    assert b"".join(packed) == gold


def test_cli_usage_and_errors(built, tmp_path):
    pmov, pm = built.build_cli()
    r = subprocess.run([pmov], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "USAGE" in r.stderr and "wrk-dir volume-id output" in r.stderr   # main.c:30-33
    r = subprocess.run([pmov, "-x", "1", "a", "0", "b"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1
    r = subprocess.run([pm], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "USAGE" in r.stderr
    # missing volume directory -> exit 1 and no output file
    out = os.path.join(str(tmp_path), "o")
    r = subprocess.run([pmov, "-k", "13", os.path.join(str(tmp_path), "nope"), "0", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and not os.path.exists(out)
    if not HAVE_GPU:
        d = util.install_golden_volumes("vols_a", tmp_path)
        r = subprocess.run([pmov, "-k", "13", d, "0", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 1 and "no usable gfx950" in r.stderr and not os.path.exists(out)


def test_new_programs_usage_and_errors(built, tmp_path):
    """oc2rm_worker / oc2asmpm: usage on bad argv (rm_one_vol_main.c:9-21, asmpm.c:3-19), exit 1 + message and no output file on a missing volume
    directory or reference and - on a machine without a GPU - on the missing device (there is no CPU fallback)"""
    built.build_cli()
    rm, asm = built.OC2RM, built.OC2ASMPM
    r = subprocess.run([rm], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "USAGE" in r.stderr and "wrk-dir reference output" in r.stderr
    r = subprocess.run([asm], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "USAGE" in r.stdout and "wrk_dir volume_id output" in r.stdout
    out = os.path.join(str(tmp_path), "o")
    r = subprocess.run([rm, "-k", "13", os.path.join(str(tmp_path), "nope"), "ref", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "ERROR" in r.stderr and not os.path.exists(out)
    r = subprocess.run([asm, "-k", "13", os.path.join(str(tmp_path), "nope"), "0", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "ERROR" in r.stderr and not os.path.exists(out)
    d = util.install_golden_volumes("vols_d", tmp_path)
    r = subprocess.run([asm, "-k", "13", d, "7", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1 and "out of range" in r.stderr
    if not HAVE_GPU:
        r = subprocess.run([asm, "-k", "13", d, "0", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 1 and "no usable gfx950" in r.stderr and not os.path.exists(out)
        e = util.install_golden_volumes("vols_e", tmp_path)
        r = subprocess.run([rm, "-k", "13", e, os.path.join(util.GOLDEN, "rm_e", "ref.vol"), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 1 and "no usable gfx950" in r.stderr and not os.path.exists(out)


def test_gapped_strings_host_helper(built):
    """necat_gapped_strings (host code of the library, no GPU): alignment columns -> the reference's two
    "ACGT-" strings.  Columns are derived from the oracle's onc_align strings; expanding them again must give
    those strings back; columns that run past a sequence are refused."""
    import numpy as np
    from necat_amd import capi
    from necat_amd.synth import _mutate
    from oracle import oracle_api as ora
    rng = np.random.default_rng(9)
    al = ora.Aligner(0.5)
    n = 0
    for it in range(12):
        g = rng.integers(0, 4, int(rng.integers(900, 4000)), dtype=np.uint8)
        q, t = _mutate(g, 0.1, rng), _mutate(g, 0.1, rng)
        qs, ts = int(0.4 * q.shape[0]), int(0.4 * t.shape[0])
        ok, qoff, qend, toff, tend, ident, qa, ta = al.align(q, qs, t, ts, 300, 4)
        assert ok and len(qa) == len(ta) > 300
        qa_b, ta_b = np.frombuffer(qa, dtype=np.uint8), np.frombuffer(ta, dtype=np.uint8)
        ops = np.where(qa_b == 45, 2, np.where(ta_b == 45, 1, np.where(qa_b == ta_b, 0, 3))).astype(np.uint8)
        packed = capi.pack_columns(ops)
        assert np.array_equal(capi.unpack_columns(packed, ops.shape[0]), ops)
        assert capi.gapped_strings(packed, ops.shape[0], q, qoff, t, toff) == (qa, ta)
        assert int((ops != 2).sum()) == qend - qoff and int((ops != 1).sum()) == tend - toff
        with pytest.raises(capi.NecatError):
            capi.gapped_strings(packed, ops.shape[0], q[:qend - 1], qoff, t, toff)
        n += 1
    al.close()
    assert n == 12 and capi.gapped_strings(np.zeros(0, dtype=np.uint8), 0, q, 0, t, 0) == (b"", b"")


def test_cns_entry_points_without_gpu(built):
    """the consensus-loop entry points: defaults as consensus/cns_options.c:10-22, argument errors reported, and no
    way around the device (a context cannot be created here, so nothing can be computed)"""
    import ctypes as C
    from necat_amd import capi
    lib = capi.load_library()
    o = capi.cns_options()
    assert (o.min_align_size, o.min_cov, o.max_cov, o.error, o.mapping_ratio, o.use_fixed_ident_cutoff) == (400, 4, 12, 0.5, 0.8, 0)
    r = C.POINTER(capi._CnsResult)()
    assert lib.necat_cns_extension_batch(None, None, None, None, None, 0, C.byref(o), C.byref(r)) == -1      # NECAT_ERR_ARG
    assert not r
    lib.necat_cns_result_free(r)                                                                             # NULL is fine
    # the role swap of oc2pcan: twice the records, the twin has query and subject exchanged
    import numpy as np
    rec = np.array([[(1 << 31) | (1 << 29) | 77, 5, 10, 20, 9, 30, 40]], dtype=np.uint32)
    both = np.frombuffer(capi.pcan_single_partition(rec.tobytes()), dtype=np.uint32).reshape(-1, 7)
    assert both.shape == (2, 7) and (both[0] == rec[0]).all()
    assert both[1].tolist() == [(1 << 30) | (1 << 29) | 77, 9, 30, 40, 5, 10, 20]

