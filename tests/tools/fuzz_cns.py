"""Randomised parity sweep of the consensus stage's extension loop (tool): random small datasets x random CnsOptions
x random speculation widths, necat_cns_extension_batch through the C ABI against the oracle's sequential loop, compared
as logs (every add_one_align call with both gapped strings, per-template cutoff / counters / ranges).

    python tests/tools/fuzz_cns.py [n_cases] [first_seed]
"""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from necat_amd import capi, synth
from oracle import oracle_api as ora
from tests import util

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    genome = int(rng.integers(20_000, 80_000))
    cov = float(rng.uniform(10, 50))
    err = float(rng.choice([0.06, 0.10, 0.13, 0.16]))
    rep = float(rng.choice([0.0, 0.0, 0.1, 0.3]))
    okw = dict(min_align_size=int(rng.choice([400, 1000, 2500])), min_cov=int(rng.choice([1, 4, 8])), max_cov=int(rng.choice([3, 8, 12, 25])),
               error=float(rng.choice([0.3, 0.5])), mapping_ratio=float(rng.choice([0.4, 0.8, 0.95])),
               use_fixed_ident_cutoff=int(rng.integers(0, 4) == 0))
    knobs = {"NECAT_CNS_SPEC": str(int(rng.choice([1, 3, 12, 50, 0]))), "NECAT_CNS_SPEC_EXTRA": str(int(rng.choice([-1, 0, 1, 7]))),
             "NECAT_BATCH": str(int(rng.choice([786432, 2048])))}
    kw = dict(util.FAST, kmer_size=int(rng.choice([12, 13, 15])), num_candidates=int(rng.choice([20, 500])))
    with tempfile.TemporaryDirectory() as td:
        multi = bool(rng.integers(0, 2))
        wrk, rs, nv = util.make_dataset(td, genome=genome, coverage=cov, seed=seed0 + case, err=err, repeat_frac=rep,
                                        vol_size=(max(300_000, int(genome * cov / 3)) if multi else synth.DEFAULT_VOL_SIZE))
        t0 = time.time()
        o = ora.options(**dict(kw, job=0, binary_output=1, num_threads=8))
        rec = b""
        for v in range(nv):
            out = os.path.join(td, "pm_%d" % v)
            ora.pm_main(o, v, wrk, out)
            rec += open(out, "rb").read()
        part = capi.pcan_single_partition(rec)
        prefix = os.path.join(td, "cands")
        util.write_partition(prefix, part)
        log = os.path.join(td, "ora.txt")
        ora.cns_run(ora.cns_options(**okw), wrk, prefix, log, full=True)
        t1 = time.time()
        os.environ.update(knobs)
        ctx = capi.Context(0)
        vol = ctx.load_merged_volumes(wrk)
        cands, off, n_all = ctx.cns_load_partition(vol, np.frombuffer(part, dtype=np.uint8))
        res = ctx.cns_extension_batch(vol, cands, off, n_all, capi.cns_options(**okw))
        roff = np.zeros(len(vol.names) + 1, dtype=np.int64)
        roff[1:] = np.cumsum(vol.sizes)
        txt = util.cns_log_text(res, cands, off, vol.codes, roff, ora.fnv64, full=True)
        stats = (int(res.n_aligned), int(res.n_used), int(res.n_rounds), int(res.overlaps.shape[0]))
        res.free(); vol.free(); ctx.close()
        ok = txt == open(log).read()
        bad += not ok
        print("case %d seed %d: genome %d cov %.0f err %.2f rep %.1f vols %d | %s %s | templates %d aligned/used/passes/overlaps %s | oracle %.1fs | %s" % (
            case, seed0 + case, genome, cov, err, rep, nv, okw, knobs, off.shape[0] - 1, stats, t1 - t0, "ok" if ok else "MISMATCH"), flush=True)
print("mismatching cases: %d of %d" % (bad, n_cases))
sys.exit(1 if bad else 0)
