"""GPU parity tests: every stage of the HIP path, called through the C ABI, against the oracle on the
same seeded inputs (bit-exact: integer / index work; ident_perc is an exactly reproducible IEEE
quotient, compared with ==)."""
import os

import numpy as np
import pytest

from tests import util
from oracle import oracle_api as ora

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(tmp_path_factory):
    d, rs, nv = util.make_dataset(tmp_path_factory.mktemp("small"), genome=150_000, coverage=18.0, seed=5)
    return d, rs


@pytest.fixture(scope="module")
def multi(tmp_path_factory):
    d, rs, nv = util.make_dataset(tmp_path_factory.mktemp("multi"), genome=200_000, coverage=20.0, seed=11, err=0.10,
                                  repeat_frac=0.05, vol_size=1_500_000)
    assert nv >= 3
    return d, rs, nv


@pytest.mark.parametrize("emit_big", [None, "1", "0"])
def test_index_matches_oracle(ctx, small, monkeypatch, emit_big):
    """k = 8: direct passes; k = 11..13: bucket partition + LDS slices (16..256 buckets); cutoffs 500 / 50 / 1.  emit_big: the slice kernel's
    instance with the big LDS ranking buffer (what a volume above 0.27 Gbp takes by itself) forced on / off."""
    if emit_big is not None:
        monkeypatch.setenv("NECAT_INDEX_EMIT_BIG", emit_big)
    d, rs = small
    vol = ctx.load_volume(os.path.join(d, "vol0"))
    for k, q in ((11, 50), (13, 500), (8, 500), (12, 1)):
        ix = ctx.build_index(vol, k, q)
        stats, offs = ix.download()
        ostats, ooffs = ora.build_index(os.path.join(d, "vol0"), k, q)
        assert np.array_equal(stats, ostats)
        assert np.array_equal(offs, ooffs)
        # the table as the device holds it (what a host-side reader copies instead of the dense array: necat_index_download_sparse)
        sp = ix.download_sparse()
        assert (sp is None) == (k < 11)
        if sp is not None:
            from necat_amd import shard
            bits, base, compact, soffs = sp
            assert np.array_equal(soffs, ooffs)
            assert int((ostats != 0).sum()) == compact.shape[0]
            h = np.concatenate([np.flatnonzero(ostats)[:5000], np.random.default_rng(k).integers(0, ostats.shape[0], 5000)]).astype(np.uint64)
            assert np.array_equal(shard.sparse_lookup(bits, base, compact, h), ostats[h.astype(np.int64)])
        ix.free()
    vol.free()


def _oracle_records(opt_kw, d, vid, tmp_path, job, binary):
    o = ora.options(**dict(opt_kw, job=job, binary_output=binary))
    out = os.path.join(str(tmp_path), "o_%d_%d_%d.out" % (vid, job, binary))
    st = ora.pm_main(o, vid, d, out)
    return out, st


@pytest.mark.parametrize("preset", ["FAST", "SENSITIVE"])
def test_candidates_match_oracle(ctx, small, tmp_path, preset):
    from necat_amd import capi
    d, rs = small
    kw = getattr(util, preset)
    out, st = _oracle_records(kw, d, 0, tmp_path, 0, 1)
    opt = capi.default_options(**dict(kw, job=0, binary_output=1))
    cands, _ = capi.pm_main(ctx, opt, 0, d)
    mine = sorted(bytes(r) for r in capi.pack_candidates(cands).astype("<u4"))
    assert len(mine) == st.n_records
    assert mine == ora.sorted_records(out, 28)
    assert len(mine) > 500


def test_m4_matches_oracle(ctx, small, tmp_path):
    from necat_amd import capi
    d, rs = small
    out, st = _oracle_records(util.FAST, d, 0, tmp_path, 1, 1)
    opt = capi.default_options(**dict(util.FAST, job=1))
    _, m4 = capi.pm_main(ctx, opt, 0, d)
    ref = np.frombuffer(open(out, "rb").read(), dtype=capi.M4_DTYPE)
    assert m4.shape[0] == ref.shape[0] == st.n_records
    assert util.m4_key_rows(m4) == util.m4_key_rows(ref)
    assert int((m4["qend"] - m4["qoff"]).sum()) == st.aligned_qbases


def test_multi_volume_all_pairs(ctx, multi, tmp_path):
    from necat_amd import capi
    d, rs, nv = multi
    kw = dict(util.SENSITIVE, kmer_size=12, kmer_cnt_cutoff=200)
    for vid in range(nv):
        out, st = _oracle_records(kw, d, vid, tmp_path, 1, 1)
        opt = capi.default_options(**dict(kw, job=1))
        _, m4 = capi.pm_main(ctx, opt, vid, d)
        ref = np.frombuffer(open(out, "rb").read(), dtype=capi.M4_DTYPE)
        assert util.m4_key_rows(m4) == util.m4_key_rows(ref), "volume %d" % vid


def test_truncation_to_num_candidates(ctx, small, tmp_path):
    """-n smaller than the per-read candidate count exercises the sort + cut of pm_worker.c:168-171."""
    from necat_amd import capi
    d, rs = small
    kw = dict(util.FAST, num_candidates=3)
    for job in (0, 1):
        out, st = _oracle_records(kw, d, 0, tmp_path, job, 1)
        opt = capi.default_options(**dict(kw, job=job))
        c, m4 = capi.pm_main(ctx, opt, 0, d)
        if job == 0:
            assert sorted(bytes(r) for r in capi.pack_candidates(c).astype("<u4")) == ora.sorted_records(out, 28)
        else:
            ref = np.frombuffer(open(out, "rb").read(), dtype=capi.M4_DTYPE)
            assert util.m4_key_rows(m4) == util.m4_key_rows(ref)


def _random_pairs(rng, n, qlo, qhi, err):
    seqs, qo, ql, to, tl = [], [], [], [], []
    pos = 0
    from necat_amd.synth import _mutate
    for _ in range(n):
        L = int(rng.integers(qlo, qhi + 1))
        t = rng.integers(0, 4, L, dtype=np.uint8)
        q = _mutate(t, err, rng)[:794]
        if q.shape[0] == 0:
            q = t[:1].copy()
        extra = rng.integers(0, 4, int(rng.integers(0, 60)), dtype=np.uint8)
        t = np.concatenate([t, extra])[:794]
        seqs += [q, t]
        qo.append(pos); ql.append(q.shape[0]); pos += q.shape[0]
        to.append(pos); tl.append(t.shape[0]); pos += t.shape[0]
    return np.concatenate(seqs), qo, ql, to, tl


@pytest.mark.parametrize("path", ["band", "recompute", "recompute_16", "recompute_rows", "recompute_quad", "recompute_fast", "recompute_fast_sorted", "recompute_fast_rolled"])
def test_edlib_blocks_match_oracle(ctx, monkeypatch, path):
    """The dominant kernel in isolation: distance, end column and the full edit path, including
    ragged sizes (1..794), failures (too divergent) and exact 512 x 512 blocks (the FULL kernel).
    band: DP passes + band records + walk; recompute: checkpoint pass + the walk that recomputes its cells (k_myers_ckg, ext_rcwalk.h; k_rcwalk3,
    ext_rcwalk3.h: a workgroup of two waves recomputes 64 blocks into 32-diagonal records, one of the two walks them column by column) at both geometries (8 words / 13 words
    per block; NECAT_RC_WW=2 = that kernel at every list size - the default takes it from 160 k blocks up); recompute_16: the same on 16-diagonal records
    (NECAT_RC3_BAND=16); recompute_rows: round 4's k_rcwalk2w (64-row records, one LDS read per walk step: NECAT_RC_WW=1 at this list size); recompute_quad:
    through k_rcwalk2 (every quad recomputes and walks its own block, NECAT_RC_WW=0); recompute_fast_sorted: the pairs of a batch in order of size, as the rounds' sorted
    list B hands them to k_myers_ckf - the blocks of a wave are then of like size and most of its 32-step windows take fast_shw_ckr's unrolled form (round 6);
    recompute_fast_rolled: NECAT_CKR_FAST=0, every window in the rolled, lane-masked form."""
    if path != "band":
        monkeypatch.setenv("NECAT_BATCH_RC", "2" if path.startswith("recompute_fast") else "1")          # 2: the checkpoint pass through k_myers_ckf (fast_shw_ckr at 8 and 16 lanes per block)
    if path == "recompute_fast_rolled":
        monkeypatch.setenv("NECAT_CKR_FAST", "0")
    # (knobs are read when a context is made: ctx.edlib_align_batch runs on a cross-check context of its own per knob environment, capi.py)
    monkeypatch.setenv("NECAT_RC_WW", {"recompute_quad": "0", "recompute_rows": "1", "recompute": "2", "recompute_16": "2"}.get(path, "1"))
    if path == "recompute_16":
        monkeypatch.setenv("NECAT_RC3_BAND", "16")
    rng = np.random.default_rng(2024)
    seqs, qo, ql, to, tl = _random_pairs(rng, 300, 1, 794, 0.15)
    # query much longer than the target: distance >= |q| - |t| > k = 0.55 * min(|q|, |t|) -> Edlib_align fails
    s2, qo2, ql2, to2, tl2 = [], [], [], [], []
    p2 = 0
    for _ in range(40):
        q = rng.integers(0, 4, int(rng.integers(500, 794)), dtype=np.uint8)
        t = rng.integers(0, 4, int(rng.integers(50, 250)), dtype=np.uint8)
        s2 += [q, t]
        qo2.append(p2); ql2.append(q.shape[0]); p2 += q.shape[0]
        to2.append(p2); tl2.append(t.shape[0]); p2 += t.shape[0]
    s2 = np.concatenate(s2)
    base = seqs.shape[0]
    # exact 512 x 512 blocks cut from longer pairs
    full_q, full_t = [], []
    from necat_amd.synth import _mutate
    for _ in range(200):
        t = rng.integers(0, 4, 700, dtype=np.uint8)
        q = _mutate(t, 0.13, rng)
        if q.shape[0] >= 512:
            full_q.append(q[:512]); full_t.append(t[:512])
    allseq = [seqs, s2] + [x for p in zip(full_q, full_t) for x in p]
    qo += [o + base for o in qo2]; to += [o + base for o in to2]; ql += ql2; tl += tl2
    pos = base + s2.shape[0]
    for _ in full_q:
        qo.append(pos); ql.append(512); pos += 512
        to.append(pos); tl.append(512); pos += 512
    allseq = np.concatenate(allseq)
    if path == "recompute_fast_sorted":
        order = sorted(range(len(qo)), key=lambda i: (tl[i], ql[i]))
        qo, ql, to, tl = ([x[i] for i in order] for x in (qo, ql, to, tl))
    dist, qend, tend, ops, ops_off = ctx.edlib_align_batch(allseq, qo, ql, to, tl, 0.5)
    nfail = 0
    for i in range(len(qo)):
        q = allseq[qo[i]:qo[i] + ql[i]]
        t = allseq[to[i]:to[i] + tl[i]]
        ok, d, qe, te, oops = ora.edlib_align(q, t, 0.5)
        if not ok:
            assert dist[i] == -1, i
            nfail += 1
            continue
        assert (dist[i], qend[i], tend[i]) == (d, qe, te), i
        assert np.array_equal(ops[ops_off[i]:ops_off[i + 1]], oops), i
    assert nfail >= 40 and nfail < len(qo) // 2
    ran = ctx._xc if ctx._xc is not None else ctx            # the context the hook ran on
    tm = ran.timings()
    assert ran.xcheck and tm.myers_word_updates > 0


def test_empty_and_tiny_inputs(ctx, tmp_path):
    """Edge cases: reads shorter than k, a volume with a single read, no candidates at all."""
    from necat_amd import capi, synth
    rng = np.random.default_rng(1)
    reads = [rng.integers(0, 4, n, dtype=np.uint8) for n in (5, 3000, 12, 4000)]
    sizes = np.array([r.shape[0] for r in reads], dtype=np.int64)
    rs = synth.ReadSet(np.concatenate(reads), np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64), sizes,
                       ["a", "b", "c", "d"])
    d = os.path.join(str(tmp_path), "tiny")
    synth.write_volume_dir(d, rs)
    opt = capi.default_options(**dict(util.FAST, job=1))
    c, m4 = capi.pm_main(ctx, opt, 0, d)
    assert c.shape[0] == 0 and m4.shape[0] == 0
    # identical reads: every later read overlaps every earlier one end to end
    base = rng.integers(0, 4, 5000, dtype=np.uint8)
    rs2 = synth.ReadSet(np.concatenate([base] * 3), np.array([0, 5000, 10000]), np.array([5000] * 3), ["x", "y", "z"])
    d2 = os.path.join(str(tmp_path), "same")
    synth.write_volume_dir(d2, rs2)
    o = ora.options(**dict(util.FAST, job=1, binary_output=1))
    out = os.path.join(str(tmp_path), "same.out")
    ora.pm_main(o, 0, d2, out)
    ref = np.frombuffer(open(out, "rb").read(), dtype=capi.M4_DTYPE)
    c, m4 = capi.pm_main(ctx, opt, 0, d2)
    assert util.m4_key_rows(m4) == util.m4_key_rows(ref)
    assert m4.shape[0] == 3 and np.all(m4["ident_perc"] == 100.0)


def test_bad_arguments_fail_loudly(ctx):
    from necat_amd import capi
    pac = np.zeros(10, dtype=np.uint8)
    with pytest.raises(capi.NecatError):
        ctx.upload_volume(pac, 40, np.array([0, 30]), np.array([20, 20]))     # gap between reads
    vol = ctx.upload_volume(pac, 40, np.array([0, 20]), np.array([20, 20]))
    with pytest.raises(capi.NecatError):
        ctx.build_index(vol, 16, 500)                                          # HashBits = 30
    vol.free()


def _fresh_ctx(threshold):
    from necat_amd import capi
    os.environ["NECAT_COOP_THRESHOLD"] = str(threshold)
    try:
        return capi.Context(0)
    finally:
        os.environ.pop("NECAT_COOP_THRESHOLD", None)


def test_coop_equals_banded(small, tmp_path):
    """The cooperative (unbanded, 8/16 lanes per block) and the lane-per-block (reference banding) DP
    kernels must give identical distances, end columns and edit paths, and identical M4 records."""
    from necat_amd import capi
    rng = np.random.default_rng(77)
    seqs, qo, ql, to, tl = _random_pairs(rng, 600, 1, 794, 0.14)
    from necat_amd.synth import _mutate
    parts = [seqs]
    pos = seqs.shape[0]
    for _ in range(300):
        t = rng.integers(0, 4, 700, dtype=np.uint8)
        q = _mutate(t, float(rng.uniform(0.02, 0.3)), rng)
        if q.shape[0] >= 512:
            parts += [q[:512], t[:512]]
            qo.append(pos); ql.append(512); pos += 512
            to.append(pos); tl.append(512); pos += 512
    allseq = np.concatenate(parts)
    outs = []
    for thr in (0, 1 << 30):
        c = _fresh_ctx(thr)
        outs.append(c.edlib_align_batch(allseq, qo, ql, to, tl, 0.5))
        d, rs = small
        opt = capi.default_options(**dict(util.FAST, job=1))
        outs[-1] = outs[-1] + (util.m4_key_rows(capi.pm_main(c, opt, 0, d)[1]),)
        c.close()
    a, b = outs
    for x, y in zip(a[:5], b[:5]):
        assert np.array_equal(x, y)
    assert a[5] == b[5] and len(a[5]) > 500
    assert int((a[0] >= 0).sum()) > 500


def test_seeding_in_several_chunks(ctx, small, tmp_path):
    """Seeding scratch is budgeted per chunk of reads; with a tiny budget (many chunks, the multi-chunk
    host assembly) and with the wave-per-strand / lane-per-strand collection kernels the candidates
    must be the very same array, in the same order."""
    from necat_amd import capi
    d, rs = small
    opt = capi.default_options(**dict(util.SENSITIVE, job=0))
    base, _ = capi.pm_main(ctx, opt, 0, d)
    assert base.shape[0] > 500
    for env in ({"NECAT_SEED_BUDGET": "30000"}, {"NECAT_SEED_WAVE": "0"}):
        os.environ.update(env)
        try:
            c = capi.Context(0)
        finally:
            for k in env:
                os.environ.pop(k, None)
        got, _ = capi.pm_main(c, opt, 0, d)
        c.close()
        assert got.tobytes() == base.tobytes(), env


@pytest.mark.parametrize("tail", [4, 1])
def test_onc_align_with_strings_matches_oracle(ctx, small, tail):
    """necat_onc_align_batch (the call the consensus stage makes, tail_match_len = 4) against the
    oracle's onc_align on every candidate: return value, coordinates, identity and both gapped strings."""
    from necat_amd import capi
    d, rs = small
    opt = capi.default_options(**dict(util.FAST, job=0))
    cands, _ = capi.pm_main(ctx, opt, 0, d)
    assert cands.shape[0] > 500
    _, _, vols = capi.load_volumes_info(d)
    vol = ctx.load_volume(vols[0][0])
    aln, ops, off = ctx.onc_align_batch(vol, vol, 0, 0, cands, opt, tail)
    vol.free()
    assert aln.shape[0] == cands.shape[0] and off.shape[0] == cands.shape[0] + 1 and int(off[-1]) == ops.shape[0]
    al = ora.Aligner(opt.error)
    n_ok = 0
    for i, c in enumerate(cands):
        q = rs.codes[rs.offsets[c["qid"]]: rs.offsets[c["qid"]] + rs.sizes[c["qid"]]]
        if c["qdir"] == 1:
            q = (3 - q[::-1]).astype(np.uint8)
        t = rs.codes[rs.offsets[c["sid"]]: rs.offsets[c["sid"]] + rs.sizes[c["sid"]]]
        ok, qoff, qend, toff, tend, ident, qa, ta = al.align(q, int(c["qoff"]), t, int(c["soff"]), opt.align_size_cutoff, tail)
        a = aln[i]
        assert (bool(a["ok"]), int(a["qoff"]), int(a["qend"]), int(a["toff"]), int(a["tend"]), int(a["align_size"])) == \
               (ok, qoff, qend, toff, tend, len(qa)), i
        assert float(a["ident_perc"]) == ident, i
        mine = capi.gapped_strings(ops[int(off[i]):int(off[i + 1])], int(a["align_size"]), q, qoff, t, toff)
        assert mine == (qa, ta), i
        n_ok += ok
    al.close()
    assert n_ok > 300


def test_onc_align_arbitrary_anchors(ctx):
    """Anchors anywhere in the overlap (left AND right extensions of several blocks), at the sequence ends,
    on unrelated sequences (the extension fails, empty alignment), both query strands, both tail lengths."""
    from necat_amd import capi
    from necat_amd.synth import _mutate, pack_2bit
    rng = np.random.default_rng(321)
    seqs, cand_rows = [], []
    for it in range(70):
        g = rng.integers(0, 4, int(rng.integers(1200, 9000)), dtype=np.uint8)
        q = _mutate(g, float(rng.uniform(0.03, 0.16)), rng)
        t = _mutate(g, float(rng.uniform(0.03, 0.16)), rng)
        if it % 7 == 3:
            t = rng.integers(0, 4, t.shape[0], dtype=np.uint8)
        qdir = it & 1
        stored_q = (3 - q[::-1]).astype(np.uint8) if qdir else q       # the volume holds the forward strand
        qid, sid = len(seqs), len(seqs) + 1
        seqs += [stored_q, t]
        for _ in range(3):
            frac = float(rng.uniform(0.0, 1.0)) if it % 5 else float(rng.integers(0, 2))
            cand_rows.append((qid, sid, qdir, int(frac * (q.shape[0] - 1)), int(frac * (t.shape[0] - 1)), q, t))
    sizes = np.array([s.shape[0] for s in seqs], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    codes = np.concatenate(seqs)
    vol = ctx.upload_volume(pack_2bit(codes), int(sizes.sum()), offs, sizes)
    cands = np.zeros(len(cand_rows), dtype=capi.CANDIDATE_DTYPE)
    for i, (qid, sid, qdir, qoff, soff, q, t) in enumerate(cand_rows):
        cands[i]["qid"], cands[i]["sid"], cands[i]["qdir"] = qid, sid, qdir
        cands[i]["qsize"], cands[i]["ssize"], cands[i]["qoff"], cands[i]["soff"] = q.shape[0], t.shape[0], qoff, soff
    opt = capi.default_options(**dict(util.FAST, job=1, align_size_cutoff=500))
    al = ora.Aligner(opt.error)
    n_ok = n_empty = 0
    for tail in (4, 1):
        aln, ops, off = ctx.onc_align_batch(vol, vol, 0, 0, cands, opt, tail)
        for i, (qid, sid, qdir, qoff, soff, q, t) in enumerate(cand_rows):
            ok, a0, a1, b0, b1, ident, qa, ta = al.align(q, qoff, t, soff, 500, tail)
            a = aln[i]
            assert (bool(a["ok"]), int(a["qoff"]), int(a["qend"]), int(a["toff"]), int(a["tend"]), int(a["align_size"]),
                    float(a["ident_perc"])) == (ok, a0, a1, b0, b1, len(qa), ident), (i, tail)
            assert capi.gapped_strings(ops[int(off[i]):int(off[i + 1])], int(a["align_size"]), q, a0, t, b0) == (qa, ta), (i, tail)
            n_ok += ok; n_empty += len(qa) == 0
    al.close(); vol.free()
    assert n_ok > 200 and n_empty > 10


def test_index_lds_slices_equal_global_atomic_passes(small):
    """Two implementations of the partitioned index build (LDS slices, the default; global-atomic bucket
    passes) must give the same arrays; k = 15 (4096 buckets x 64 slices, the bench configuration) is too big
    for the CPU oracle in a test, so it is checked this way and end-to-end by the E. coli golden records."""
    from necat_amd import capi
    d, rs = small
    out = []
    for env in ("1", "0"):
        os.environ["NECAT_INDEX_LDS"] = env
        try:
            c = capi.Context(0)
        finally:
            os.environ.pop("NECAT_INDEX_LDS", None)
        vol = c.load_volume(os.path.join(d, "vol0"))
        ix = c.build_index(vol, 15, 500)
        out.append(ix.download())
        ix.free(); vol.free(); c.close()
    assert np.array_equal(out[0][1], out[1][1]) and out[0][1].shape[0] > 1_000_000
    assert np.array_equal(out[0][0], out[1][0])


def test_map_pair_equals_find_then_extend(ctx, multi):
    """necat_map_pair (candidates never leave the device) = necat_find_candidates + necat_extend, for the self
    pair and a cross-volume pair."""
    from necat_amd import capi
    d, rs, nv = multi
    _, _, vols = capi.load_volumes_info(d)
    opt = capi.default_options(**dict(util.SENSITIVE, job=1))
    ref = ctx.load_volume(vols[0][0])
    ix = ctx.build_index(ref, opt.kmer_size, opt.kmer_cnt_cutoff)
    total = 0
    for i in (0, 1):
        reads = ref if i == 0 else ctx.load_volume(vols[i][0])
        c = ctx.find_candidates(ix, ref, reads, vols[i][1], vols[0][1], opt, True)
        a = ctx.extend(ref, reads, vols[i][1], vols[0][1], c, opt, 1)
        b, nc = ctx.map_pair(ix, ref, reads, vols[i][1], vols[0][1], opt, True, 1)
        assert nc == c.shape[0] and util.m4_key_rows(a) == util.m4_key_rows(b)
        total += b.shape[0]
        if reads is not ref:
            reads.free()
    ix.free(); ref.free()
    assert total > 300


@pytest.mark.parametrize("lanes", [dict(), dict(NECAT_EXT_OVERLAP="0"), dict(NECAT_EXT_OVERLAP_PCT="100"), dict(NECAT_EXT_OVERLAP_PCT="0"),
                                   dict(NECAT_BATCH="100000", NECAT_EXT_OVERLAP_MIN="1000", NECAT_EXT_OVERLAP_SPLIT="30"),
                                   dict(NECAT_EXT_LANES="3"), dict(NECAT_EXT_LANES="4", NECAT_EXT_OVERLAP_PCT="0"), dict(NECAT_EXT_LANES="1")],
                         ids=["overlap", "one_lane", "overlap_at_once", "overlap_never_early", "one_batch_cut_in_two", "three_lanes", "four_lanes_never_early", "lanes_1"])
def test_extension_in_several_batches(ctx, small, lanes):
    """Candidates go through the extension in batches (NECAT_BATCH, default 786 432); tiny batches - many batch
    switches, lists far below the single-pass threshold - must give the same M4 records and the same alignments.
    With two lanes (the default: batch i + 1's first rounds beside batch i's last, stage_extend.inl extend_impl), with one, with three and four (NECAT_EXT_LANES);
    the next batch started as soon as a lane is free / only when the previous one has ended; one batch cut in two uneven halves."""
    from necat_amd import capi
    d, rs = small
    opt = capi.default_options(**dict(util.FAST, job=1))
    _, base = capi.pm_main(ctx, opt, 0, d)
    env = dict({"NECAT_BATCH": "320"}, **lanes)
    os.environ.update(env)
    try:
        c = capi.Context(0)
    finally:
        for k_ in env:
            os.environ.pop(k_, None)
    _, got = capi.pm_main(c, opt, 0, d)
    cands, _ = capi.pm_main(c, capi.default_options(**dict(util.FAST, job=0)), 0, d)
    _, _, vols = capi.load_volumes_info(d)
    vol = c.load_volume(vols[0][0])
    a1 = c.onc_align_batch(vol, vol, 0, 0, cands[:900], opt, 4)
    vol.free(); c.close()
    vol = ctx.load_volume(vols[0][0])
    a0 = ctx.onc_align_batch(vol, vol, 0, 0, cands[:900], opt, 4)
    vol.free()
    assert util.m4_key_rows(got) == util.m4_key_rows(base) and base.shape[0] > 500
    for x, y in zip(a0, a1):
        assert x.tobytes() == y.tobytes()


def test_product_library_refuses_cross_check_knobs(small, monkeypatch):
    """libnecat_hip.so is built without the kernel families its default paths replaced (necat_hip.hip, NECAT_BUILD_CROSSCHECK): a knob that selects one of them
    fails the call and says so - no other path runs in its place; the cross-check build takes the same knob (the alternative-path tests)"""
    from necat_amd import capi
    d, rs = small
    o1 = capi.default_options(**dict(util.FAST, job=1))
    monkeypatch.setenv("NECAT_RCWALK", "0")
    assert capi.needs_xcheck()
    c = capi.Context(0, xcheck=False)
    try:
        with pytest.raises(capi.NecatError, match="cross-check"):
            capi.pm_main(c, o1, 0, d)
    finally:
        c.close()
    x = capi.Context(0)                      # (auto: the cross-check build)
    try:
        assert x.xcheck and capi.pm_main(x, o1, 0, d)[1].shape[0] > 500
    finally:
        x.close()
    monkeypatch.setenv("NECAT_RCWALK", "512")
    monkeypatch.setenv("NECAT_RC_WW", "0")
    with pytest.raises(capi.NecatError):
        capi.Context(0, xcheck=False)        # k_rcwalk2 is chosen inside a launcher that cannot fail: refused when the context is made


def test_capped_band_pool_runs_lists_in_chunks(ctx, small, tmp_path, monkeypatch):
    """NECAT_BAND_POOL_MB (set by the command-line programs: a fresh process pays for every GB of VRAM it touches) caps the
    band-record pools; a round's list then runs as several DP + walk launches over the same pool.  An 8 MB cap = chunks of
    128 list-A blocks / 64 list-B blocks: same M4 records as the uncapped run, and the short-read case (long list-B chains)
    against the oracle."""
    from necat_amd import capi
    d, rs = small
    opt = capi.default_options(**dict(util.FAST, job=1))
    _, base = capi.pm_main(ctx, opt, 0, d)
    monkeypatch.setenv("NECAT_BAND_POOL_MB", "8")
    c = capi.Context(0)
    try:
        _, got = capi.pm_main(c, opt, 0, d)
        assert util.m4_key_rows(got) == util.m4_key_rows(base) and base.shape[0] > 500
        # the alignment-keeping mode (columns exported by the walk) through the same chunks
        cands, _ = capi.pm_main(c, capi.default_options(**dict(util.FAST, job=0)), 0, d)
        _, _, vols = capi.load_volumes_info(d)
        v1 = c.load_volume(vols[0][0]); a1 = c.onc_align_batch(v1, v1, 0, 0, cands[:1500], opt, 4); v1.free()
        v0 = ctx.load_volume(vols[0][0]); a0 = ctx.onc_align_batch(v0, v0, 0, 0, cands[:1500], opt, 4); v0.free()
        for x, y in zip(a0, a1):
            assert x.tobytes() == y.tobytes()
        kw = dict(util.FAST, kmer_size=12, align_size_cutoff=400, num_threads=4)
        d2, rs2, nv = util.make_dataset(tmp_path, genome=60_000, coverage=40.0, seed=43, err=0.10, mean_len=1350.0, sd_len=200.0, min_len=1000)
        out, st = _oracle_records(kw, d2, 0, tmp_path, 1, 1)
        ref = np.frombuffer(open(out, "rb").read(), dtype=capi.M4_DTYPE)
        _, m4 = capi.pm_main(c, capi.default_options(**dict(kw, job=1)), 0, d2)
        assert ref.shape[0] > 2000 and util.m4_key_rows(m4) == util.m4_key_rows(ref)
    finally:
        c.close()


@pytest.mark.parametrize("knob", ["NECAT_CHAIN_WAVE=0", "NECAT_RCWALK=0 NECAT_FAST16=1", "NECAT_RCWALK=0 NECAT_WALK=1", "NECAT_FAST=0", "NECAT_SEED_WAVE=0",
                                  "NECAT_TAIL_FUSED=0", "NECAT_TAIL_FUSED=100000000", "NECAT_RCWALK=0 NECAT_WALK_WAVE=0", "NECAT_RCWALK=0 NECAT_WALK_WAVE=100000000 NECAT_TAIL_FUSED=0", "NECAT_SEED_KST=0",
                                  "NECAT_RCWALK=0 NECAT_RC_LISTB=0 NECAT_TAIL_FUSED=0",
                                  "NECAT_RCWALK=0", "NECAT_RCWALK=1 NECAT_TAIL_FUSED=0 NECAT_WALK_WAVE=0", "NECAT_RCWALK=1 NECAT_RC_MAXDIST=90 NECAT_TAIL_FUSED=0",
                                  "NECAT_RCWALK=1 NECAT_RC_CARRY=0 NECAT_TAIL_FUSED=0", "NECAT_RCWALK=1 NECAT_RC_CARRY=0 NECAT_RC_MAXDIST=90",
                                  "NECAT_RCWALK=1 NECAT_RC_POOL_MB=1", "NECAT_RCWALK=1 NECAT_RC_RAGGED=0 NECAT_TAIL_FUSED=0",
                                  "NECAT_RC_LISTB=0", "NECAT_RC_LISTB=1 NECAT_TAIL_FUSED=0 NECAT_RC_POOL_MB=1",
                                  "NECAT_RC_WW=0 NECAT_RCWALK=1 NECAT_TAIL_FUSED=0", "NECAT_RC_WW=1 NECAT_RCWALK=1 NECAT_TAIL_FUSED=0",
                                  "NECAT_RC_WW=2 NECAT_RCWALK=1 NECAT_TAIL_FUSED=0", "NECAT_RC_WW=2 NECAT_RC3_BAND=16 NECAT_RCWALK=1 NECAT_TAIL_FUSED=0", "NECAT_RC3_MIN=64 NECAT_RCWALK=1 NECAT_TAIL_FUSED=0", "NECAT_RC_WW=1 NECAT_RC_PREFETCH=1 NECAT_RCWALK=1 NECAT_TAIL_FUSED=0", "NECAT_RC_MERGE=0 NECAT_RCWALK=1 NECAT_TAIL_FUSED=0",
                                  "NECAT_RC_MERGE=1 NECAT_RCWALK=1 NECAT_TAIL_FUSED=0 NECAT_RC_POOL_MB=1", "NECAT_RC_FASTB=0 NECAT_TAIL_FUSED=0",
                                  "NECAT_CK_POST=0 NECAT_RCWALK=1 NECAT_TAIL_FUSED=0", "NECAT_CKR_FAST=0 NECAT_RCWALK=1 NECAT_TAIL_FUSED=0",
                                  "NECAT_RC_PIPE=3 NECAT_RC_PIPE_MIN=64 NECAT_RCWALK=1 NECAT_TAIL_FUSED=0", "NECAT_RC_PRIO=6 NECAT_RCWALK=1 NECAT_TAIL_FUSED=0"])
def test_alternative_kernel_paths_give_the_same_records(ctx, small, monkeypatch, knob):
    """Code paths kept behind a knob (the lane-0 chain DP, the 16-block / 4-lane NW kernel, the restated walk, the general DP
    path without the full-block fast path, lane-per-strand seed collection, every round / no round through the one-launch
    LDS-band kernel of the late rounds, every list / no list walked by one wave per block) must stay correct: same candidates and same M4 records as the default paths (which the
    other tests pin to the oracle)."""
    from necat_amd import capi
    d, rs = small
    o1 = capi.default_options(**dict(util.FAST, job=1))
    o0 = capi.default_options(**dict(util.FAST, job=0))
    c_base, _ = capi.pm_main(ctx, o0, 0, d)
    _, m_base = capi.pm_main(ctx, o1, 0, d)
    for kv in knob.split():
        name, val = kv.split("=")
        monkeypatch.setenv(name, val)
    c = capi.Context(0)          # knobs are read when a context is created
    try:
        c_got, _ = capi.pm_main(c, o0, 0, d)
        _, m_got = capi.pm_main(c, o1, 0, d)
    finally:
        c.close()
        monkeypatch.undo()
    assert c_got.tobytes() == c_base.tobytes() and c_base.shape[0] > 500
    assert util.m4_key_rows(m_got) == util.m4_key_rows(m_base) and m_base.shape[0] > 500


@pytest.mark.parametrize("knob", [None, "NECAT_CHAIN_WAVE=0"])
def test_long_chains_on_both_strands(ctx, tmp_path, monkeypatch, knob):
    """Low error, high coverage, k = 11, z = 5: evaluations with more than 256 co-linear seeds (the chain scratch in global memory
    instead of LDS) on BOTH strands of a read, which two waves evaluate at the same time.  Found by tests/tools/fuzz_parity.py
    (seed 7015): the strands used to share one chain scratch per read and a chain of one strand could surface under the other.
    Also with the lane-0 chain DP, which keeps every evaluation's seeds in that scratch."""
    from necat_amd import capi
    kw = dict(kmer_size=11, scan_window=5, kmer_cnt_cutoff=20, block_size=2000, block_score_cutoff=3, num_candidates=500,
              align_size_cutoff=1000, ddfs_cutoff=0.25, error=0.3, num_output=500, num_threads=2, use_hdr_as_id=0)
    d, rs, nv = util.make_dataset(tmp_path, genome=173352, coverage=25.2, seed=7015, err=0.04, repeat_frac=0.0)
    o = ora.options(**dict(kw, job=0, binary_output=1))
    out = os.path.join(str(tmp_path), "o.out")
    ora.pm_main(o, 0, d, out)
    want = ora.sorted_records(out, 28)
    assert len(want) > 5000
    if knob:
        monkeypatch.setenv(*knob.split("="))
    c = capi.Context(0)
    try:
        for it in range(3):
            cands, _ = capi.pm_main(c, capi.default_options(**dict(kw, job=0, binary_output=1)), 0, d)
            got = sorted(bytes(r) for r in capi.pack_candidates(cands).astype("<u4"))
            assert got == want, "run %d" % it
            assert int(cands["score"].max()) > 256          # such evaluations exist
    finally:
        c.close()


def test_ultra_long_reads(ctx, tmp_path):
    """reads of 60-200 kb (hundreds of 512-bp blocks per alignment, hundreds of extension rounds, long chains in the
    seeding stage): candidates, M4 records and the alignments with their strings equal the oracle's"""
    from necat_amd import capi
    d, rs, nv = util.make_dataset(tmp_path, genome=260_000, coverage=5.0, seed=41, err=0.12, mean_len=130_000.0, sd_len=40_000.0,
                                  min_len=60_000)
    assert int(rs.sizes.max()) > 150_000 and nv == 1
    kw = dict(util.FAST, kmer_size=13)
    for job in (0, 1):
        o = ora.options(**dict(kw, job=job, binary_output=1))
        out = os.path.join(str(tmp_path), "o%d.out" % job)
        ora.pm_main(o, 0, d, out)
        cands, m4 = capi.pm_main(ctx, capi.default_options(**dict(kw, job=job, binary_output=1)), 0, d)
        if job == 0:
            assert sorted(bytes(r) for r in capi.pack_candidates(cands).astype("<u4")) == ora.sorted_records(out, 28)
            assert cands.shape[0] > 5
            keep = cands
        else:
            ref = np.frombuffer(open(out, "rb").read(), dtype=capi.M4_DTYPE)
            assert util.m4_key_rows(m4) == util.m4_key_rows(ref) and m4.shape[0] > 3
            assert int((ref["qend"] - ref["qoff"]).max()) > 60_000
    vol = ctx.load_volume(os.path.join(d, "vol0"))
    opt = capi.default_options(**dict(kw, job=1))
    sel = keep[:12]
    aln, ops, off = ctx.onc_align_batch(vol, vol, 0, 0, sel, opt, 4)
    al = ora.Aligner(opt.error)
    for i, c in enumerate(sel):
        q = rs.codes[rs.offsets[c["qid"]]: rs.offsets[c["qid"]] + rs.sizes[c["qid"]]]
        if c["qdir"] == 1:
            q = (3 - q[::-1]).astype(np.uint8)
        t = rs.codes[rs.offsets[c["sid"]]: rs.offsets[c["sid"]] + rs.sizes[c["sid"]]]
        ok, a0, a1, b0, b1, ident, qa, ta = al.align(q, int(c["qoff"]), t, int(c["soff"]), opt.align_size_cutoff, 4)
        a = aln[i]
        assert (bool(a["ok"]), int(a["qoff"]), int(a["qend"]), int(a["toff"]), int(a["tend"]), float(a["ident_perc"])) == (ok, a0, a1, b0, b1, ident), i
        assert capi.gapped_strings(ops[int(off[i]):int(off[i + 1])], int(a["align_size"]), q, a0, t, b0) == (qa, ta), i
    al.close(); vol.free()



def test_short_reads_list_b_chains_repeatable(ctx, tmp_path, monkeypatch):
    """Reads of 1.0 - 1.7 kb: an overlap is short on both sides of its anchor, so a left LAST block bigger than 512 (list
    B) is followed by a right block that is also last and bigger than 512 (list B again) - the case in which, with three
    list buffers, round r's list-B kernels appended into the buffer round r - 1's list-B kernels were still reading.
    Unsorted list B (NECAT_SORT_B=0: the kernels read the very buffer that receives appends), several runs, every run
    against the oracle."""
    from necat_amd import capi
    monkeypatch.setenv("NECAT_SORT_B", "0")
    kw = dict(util.FAST, kmer_size=12, align_size_cutoff=400, num_threads=4)
    d, rs, nv = util.make_dataset(tmp_path, genome=60_000, coverage=40.0, seed=41, err=0.10, mean_len=1350.0, sd_len=200.0, min_len=1000)
    out, st = _oracle_records(kw, d, 0, tmp_path, 1, 1)
    ref = np.frombuffer(open(out, "rb").read(), dtype=capi.M4_DTYPE)
    assert ref.shape[0] > 2000
    want = util.m4_key_rows(ref)
    c2 = capi.Context(0)          # knobs are read when a context is created
    try:
        opt = capi.default_options(**dict(kw, job=1))
        for it in range(6):
            _, m4 = capi.pm_main(c2, opt, 0, d)
            assert util.m4_key_rows(m4) == want, "run %d differs from the oracle" % it
        assert c2.timings().rounds >= 3
    finally:
        c2.close()
