"""necat_asm_align_batch through the C ABI against the oracle's onc_align at block size 2048 / tail match length 8 (= asm_pm/blockwise_edlib.c, which
tests/test_asmpm.py pins to the reference's own oc2asmpm): anchors anywhere in the overlap, at the sequence ends, on unrelated sequences, subjects on
both strands, reads of corrected and of raw quality."""
import numpy as np
import pytest

from necat_amd import capi
from necat_amd.synth import _mutate, pack_2bit
from oracle import oracle_api as ora

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("impl", ["cooperative", "band", "lane"])
def test_asm_align_arbitrary_anchors(ctx, monkeypatch, impl):
    """cooperative: the default path (asm_coop.h - the extension stage's round structure at the 2048-bp geometry, blocks through the
    checkpoint pass + recomputing walk of ext_rcwalk.h); band: the same rounds through the two-pass kernel + band records + wave walk
    (NECAT_ASM_RC=0); lane: the lane-per-alignment kernel they replaced, kept as a further implementation (NECAT_ASM_LANE=1)"""
    own = None
    if impl != "cooperative":
        monkeypatch.setenv("NECAT_ASM_LANE" if impl == "lane" else "NECAT_ASM_RC", "1" if impl == "lane" else "0")
        own = ctx = capi.Context(0)          # knobs are read when a context is created
    rng = np.random.default_rng(654)
    seqs, rows = [], []
    for it in range(60):
        g = rng.integers(0, 4, int(rng.integers(1500, 12000)), dtype=np.uint8)
        e = float(rng.uniform(0.005, 0.06)) if it % 4 else float(rng.uniform(0.10, 0.16))
        q = _mutate(g, e, rng)
        t = _mutate(g, e, rng)
        if it % 9 == 4:
            t = rng.integers(0, 4, t.shape[0], dtype=np.uint8)
        sdir = it & 1
        stored_t = (3 - t[::-1]).astype(np.uint8) if sdir else t       # the volume holds the forward strand; the alignment sees strand sdir = t
        qid, sid = len(seqs), len(seqs) + 1
        seqs += [q, stored_t]
        for _ in range(3):
            frac = float(rng.uniform(0.0, 1.0)) if it % 5 else float(rng.integers(0, 2))
            rows.append((qid, sid, sdir, int(frac * (q.shape[0] - 1)), int(frac * (t.shape[0] - 1)), q, t))
    sizes = np.array([s.shape[0] for s in seqs], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    vol = ctx.upload_volume(pack_2bit(np.concatenate(seqs)), int(sizes.sum()), offs, sizes)
    anchors = np.zeros(len(rows), dtype=capi.ASM_ANCHOR_DTYPE)
    for i, (qid, sid, sdir, qoff, soff, q, t) in enumerate(rows):
        anchors[i] = (qid, sid, sdir, qoff, soff)
    aln, ops, off = ctx.asm_align_batch(vol, vol, 0, 0, anchors, 0.5, 400)
    al = ora.Aligner(0.5)
    n_ok = n_empty = 0
    for i, (qid, sid, sdir, qoff, soff, q, t) in enumerate(rows):
        ok, a0, a1, b0, b1, ident, qa, ta = al.align(q, qoff, t, soff, 400, 8, block_size=2048)
        a = aln[i]
        got = (bool(a["ok"]), int(a["qoff"]), int(a["qend"]), int(a["toff"]), int(a["tend"]), int(a["align_size"]), float(a["ident_perc"]))
        assert got == (ok, a0, a1, b0, b1, len(qa), ident), i
        assert capi.gapped_strings(ops[int(off[i]):int(off[i + 1])], int(a["align_size"]), q, a0, t, b0) == (qa, ta), i
        n_ok += ok
        n_empty += len(qa) == 0
    al.close()
    assert n_ok > 100 and n_empty > 5
    # arguments: a strand that is neither 0 nor 1, an anchor outside its read, no anchors at all
    bad = anchors[:1].copy()
    bad["sdir"] = 2
    with pytest.raises(capi.NecatError):
        ctx.asm_align_batch(vol, vol, 0, 0, bad)
    bad = anchors[:1].copy()
    bad["qoff"] = 10 ** 8
    with pytest.raises(capi.NecatError):
        ctx.asm_align_batch(vol, vol, 0, 0, bad)
    aln, ops, off = ctx.asm_align_batch(vol, vol, 0, 0, anchors[:0])
    assert aln.shape[0] == 0 and off.shape[0] == 1
    if impl != "lane":
        tm = ctx.timings()
        assert tm.rounds >= 1 or len(rows) == 0          # (the empty call above ran no round; the counters are the last non-empty call's only if > 0)
    vol.free()
    if own is not None:
        own.close()
