"""GPU tests of the process boundary: the oc2pmov / oc2pm programs, run exactly as necat.pl runs them
(necat.pl:197), must write record files that are byte-identical - after sorting records - to what the
REFERENCE wrote for the same volumes and flags (tests/golden, generated from oracle/_ref)."""
import hashlib
import json
import os
import subprocess

import pytest

from tests import util
from oracle import oracle_api as ora

pytestmark = pytest.mark.gpu
MANIFEST = json.load(open(os.path.join(util.GOLDEN, "manifest.json")))


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_oc2pmov_reproduces_reference_records(name, tmp_path, built):
    pmov, _ = built.build_cli()
    m = MANIFEST[name]
    d = util.install_golden_volumes(m["dataset"], tmp_path)
    o = ora.options(**m["options"])
    out = os.path.join(str(tmp_path), "pm_result")
    r = subprocess.run([pmov] + ora.opt_argv(o) + [d, str(m["vid"]), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert "'pairwise mapping v%d vs v%d' takes" % (m["vid"], m["vid"]) in r.stdout
    recs = ora.sorted_records(out, ora.record_size(o))
    assert len(recs) == m["records"]
    assert hashlib.md5(b"".join(recs)).hexdigest() == m["md5"]
    assert not os.path.exists(out + ".part")


@pytest.mark.parametrize("gpus", [None, "0,0"])
def test_oc2pm_wrapper_concatenates_volumes(tmp_path, built, gpus):
    """gpus = "0,0": two oc2pmov children at a time (one per listed device - here the same one twice),
    the multi-GPU scheduling of reference volumes."""
    pmov, pm = built.build_cli()
    d = util.install_golden_volumes("vols_b", tmp_path)
    o = ora.options(**MANIFEST["b_v0_m4_txt"]["options"])
    out = os.path.join(str(tmp_path), "all.m4")
    env = dict(os.environ)
    if gpus:
        env["NECAT_GPUS"] = gpus
    r = subprocess.run([pm] + ora.opt_argv(o) + [d, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert r.returncode == 0, r.stderr
    got = sorted(open(out, "rb").read().splitlines(keepends=True))
    want = []
    for vid in range(3):
        p = os.path.join(str(tmp_path), "o%d" % vid)
        ora.pm_main(o, vid, d, p)
        want += open(p, "rb").read().splitlines(keepends=True)
    assert got == sorted(want)
    for vid in range(3):       # pairwise_mapping/main.c:55-70, :104-112
        assert os.path.exists(os.path.join(d, "pm%d.finished" % vid))
        assert not os.path.exists(os.path.join(d, "pm_result_%d" % vid))


def test_oc2pm_worker_failure_is_reported(tmp_path, built):
    """a volume job that fails (here: an unreadable volume file) makes oc2pm exit 1, leaves no pm<i>.finished for that volume and no
    merged output - the grid driver re-queues the script (pairwise_mapping/main.c:95-112 checks the child's exit status)"""
    pmov, pm = built.build_cli()
    d = util.install_golden_volumes("vols_b", tmp_path)
    o = ora.options(**MANIFEST["b_v0_m4_txt"]["options"])
    nv, nr, vols = __import__("necat_amd.capi", fromlist=["x"]).load_volumes_info(d)
    os.truncate(vols[2][0], 40)                     # volume 2 is needed by every job
    out = os.path.join(str(tmp_path), "all.m4")
    r = subprocess.run([pm] + ora.opt_argv(o) + [d, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 1
    assert "ERROR" in r.stderr
    assert not os.path.exists(out)
    assert not any(os.path.exists(os.path.join(d, "pm%d.finished" % v)) for v in range(3))


@pytest.mark.parametrize("job", [0, 1])
def test_pair_lanes_write_the_same_file(tmp_path, built, job):
    """NECAT_PAIR_LANES (pm_job.h): a job's query volumes mapped on 1, 2 or 3 contexts of the device side by side - the job's file is written in unit order
    whatever lane a unit ran on, so it is the same BYTES in every lane mode (and the oracle's records, sorted); the log names the pairs in the reference's order"""
    pmov, _ = built.build_cli()
    d = util.install_golden_volumes("vols_b", tmp_path)
    o = ora.options(**dict(MANIFEST["b_v0_m4_txt"]["options"], job=job))
    files = {}
    for lanes in (1, 2, 3):
        out = os.path.join(str(tmp_path), "pm_result_l%d" % lanes)
        r = subprocess.run([pmov] + ora.opt_argv(o) + [d, "0", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, NECAT_PAIR_LANES=str(lanes)))
        assert r.returncode == 0, r.stderr
        at = [r.stdout.index("'pairwise mapping v%d vs v0' takes" % q) for q in range(3)]
        assert at == sorted(at)
        files[lanes] = open(out, "rb").read()
        assert not os.path.exists(out + ".part")
    assert len(files[1]) > 0
    if job == 0:
        assert files[1] == files[2] == files[3]
    else:
        # (the M4 records of ONE pair leave the device in the order its extension batches end - compared sorted, like everywhere; across pairs the order is the units')
        from necat_amd import capi
        _, _, vols = capi.load_volumes_info(d)
        starts = [v[1] for v in vols]              # volume_names.txt: name, first read id, reads
        for lanes in (1, 2, 3):
            assert sorted(files[lanes].splitlines()) == sorted(files[1].splitlines())
            qvol = [max(k for k, s0 in enumerate(starts) if int(ln.split(b"\t")[0]) >= s0) for ln in files[lanes].splitlines()]
            assert qvol == sorted(qvol) and set(qvol) == {0, 1, 2}
    want = os.path.join(str(tmp_path), "oracle")
    ora.pm_main(o, 0, d, want)
    assert sorted(files[1].splitlines(keepends=True)) == sorted(open(want, "rb").read().splitlines(keepends=True))


def test_contexts_on_threads_give_the_same_records(tmp_path):
    """several contexts of one device used from several host threads at once (bench.py's steps in flight, the programs' pair lanes): every call returns what it
    returns alone - index, candidates and M4 records compared with a context that ran by itself"""
    import threading
    import numpy as np
    from necat_amd import capi, synth
    rs = synth.simulate_reads(300_000, 20.0, seed=5)
    pac = synth.pack_2bit(rs.codes)
    kw = dict(kmer_size=13, scan_window=10, kmer_cnt_cutoff=500, block_size=2000, block_score_cutoff=3, num_candidates=500, align_size_cutoff=1000, error=0.5, use_hdr_as_id=0)
    opt = capi.default_options(**dict(kw, job=1))
    ctxs = [capi.Context(0) for _ in range(3)]
    vol = ctxs[0].upload_volume(pac, rs.nbases, rs.offsets, rs.sizes)          # ONE resident volume, mapped by every context

    def one(c):
        ix = c.build_index(vol, opt.kmer_size, opt.kmer_cnt_cutoff)
        m4, nc = c.map_pair(ix, vol, vol, 0, 0, opt, True, 1)
        ix.free()
        return nc, np.sort(m4.copy(), order=["qid", "sid", "qoff", "soff", "qend", "send"])
    nc0, want = one(ctxs[0])
    assert want.shape[0] > 1000
    got, errs = {}, []

    def work(i):
        try:
            for rep in range(4):
                got[(i, rep)] = one(ctxs[i])
        except BaseException as e:
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert len(got) == 12
    for nc, m4 in got.values():
        assert nc == nc0 and np.array_equal(m4, want)
    vol.free()
    for c in ctxs:
        c.close()
