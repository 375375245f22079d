"""Randomised parity sweep (tool): random small datasets x random option sets, the HIP path through the C ABI
against the oracle - candidates (-j 0, packed 28-byte records) and M4 records (-j 1), every volume.

    python tests/tools/fuzz_parity.py [n_cases] [first_seed] [knobs]

knobs = 1: every case runs in a context of its own with random scheduling knobs (band-pool cap, candidates per batch, seeding
scratch budget: lists in chunks, several batches, several seeding chunks) - the records must not depend on them.
"""
import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from necat_amd import capi, synth
from oracle import oracle_api as ora
from tests import util

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
knobs = len(sys.argv) > 3 and sys.argv[3] in ("1", "2")
ctx = None if knobs else capi.Context(0)
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    if knobs:
        krng = np.random.default_rng(77 + seed0 + case)
        env = {"NECAT_BAND_POOL_MB": str(int(krng.choice([4, 16, 64, 16384]))), "NECAT_BATCH": str(int(krng.choice([192, 1024, 786432]))),
               "NECAT_SEED_BUDGET": str(int(krng.choice([20000, 200000, 48000000]))), "NECAT_SINGLE_PASS": str(int(krng.choice([0, 64, 4096]))),
               "NECAT_RC_POOL_MB": str(int(krng.choice([1, 8, 2048]))), "NECAT_RCWALK": str(int(krng.choice([1, 512, 4096]))), "NECAT_TAIL_FUSED": str(int(krng.choice([0, 512]))),
               "NECAT_RC_LISTB": str(int(krng.choice([0, 1, 1]))), "NECAT_RC_RAGGED": str(int(krng.choice([0, 1, 1])))}
        os.environ.update(env)
        if ctx is not None:
            ctx.close()
        ctx = capi.Context(0)
    genome = int(rng.integers(40_000, 220_000))
    cov = float(rng.uniform(8, 30))
    err = float(rng.choice([0.04, 0.08, 0.12, 0.15]))
    rep = float(rng.choice([0.0, 0.0, 0.05, 0.2]))
    multi = bool(rng.integers(0, 3) == 0)
    # knobs = 2: also random read-length distributions (short reads: list-B chains; long ones: many rounds, chains of > 256 seeds)
    lens = {}
    if len(sys.argv) > 3 and sys.argv[3] == "2":
        ml = float(rng.choice([1400.0, 3000.0, 8000.0, 20000.0]))
        lens = dict(mean_len=ml, sd_len=ml * float(rng.choice([0.1, 0.4])), min_len=int(max(1000, ml * 0.3)))
    kw = dict(kmer_size=int(rng.choice([11, 12, 13, 14])), scan_window=int(rng.choice([5, 10, 20])),
              kmer_cnt_cutoff=int(rng.choice([20, 100, 500])), block_size=int(rng.choice([1000, 2000, 3000])),
              block_score_cutoff=int(rng.choice([2, 3, 4])), num_candidates=int(rng.choice([3, 30, 500])),
              align_size_cutoff=int(rng.choice([500, 1000, 2000])), ddfs_cutoff=0.25, error=float(rng.choice([0.3, 0.5])),
              num_output=500, num_threads=2, use_hdr_as_id=0)
    with tempfile.TemporaryDirectory() as td:
        d, rs, nv = util.make_dataset(td, genome=genome, coverage=cov, seed=seed0 + case, err=err, repeat_frac=rep,
                                      vol_size=(max(300_000, int(genome * cov / 3)) if multi else synth.DEFAULT_VOL_SIZE), **lens)
        t0 = time.time()
        bad0 = bad
        for vid in range(nv):
            for job in (0, 1):
                o = ora.options(**dict(kw, job=job, binary_output=1))
                out = os.path.join(td, "o.out")
                st = ora.pm_main(o, vid, d, out)
                opt = capi.default_options(**dict(kw, job=job, binary_output=1))
                cands, m4 = capi.pm_main(ctx, opt, vid, d)
                if job == 0:
                    mine = sorted(bytes(r) for r in capi.pack_candidates(cands).astype("<u4"))
                    ok = mine == ora.sorted_records(out, 28)
                    nrec = len(mine)
                else:
                    ref = np.frombuffer(open(out, "rb").read(), dtype=capi.M4_DTYPE)
                    ok = util.m4_key_rows(m4) == util.m4_key_rows(ref)
                    nrec = m4.shape[0]
                if job == 0 and nv == 1 and cands.shape[0]:
                    # the consensus stage's aligner call on a sample of the candidates: strings vs the oracle's onc_align
                    sel = cands[:: max(1, cands.shape[0] // 200)][:200]
                    vol = ctx.load_volume(capi.load_volumes_info(d)[2][0][0])
                    aln, ops, off = ctx.onc_align_batch(vol, vol, 0, 0, sel, opt, 4)
                    vol.free()
                    al = ora.Aligner(opt.error)
                    for i, cnd in enumerate(sel):
                        q = rs.codes[rs.offsets[cnd["qid"]]: rs.offsets[cnd["qid"]] + rs.sizes[cnd["qid"]]]
                        if cnd["qdir"] == 1:
                            q = (3 - q[::-1]).astype(np.uint8)
                        t = rs.codes[rs.offsets[cnd["sid"]]: rs.offsets[cnd["sid"]] + rs.sizes[cnd["sid"]]]
                        r = al.align(q, int(cnd["qoff"]), t, int(cnd["soff"]), opt.align_size_cutoff, 4)
                        a = aln[i]
                        mine = (bool(a["ok"]), int(a["qoff"]), int(a["qend"]), int(a["toff"]), int(a["tend"]), float(a["ident_perc"]))
                        strs = capi.gapped_strings(ops[int(off[i]):int(off[i + 1])], int(a["align_size"]), q, r[1], t, r[3]) if mine[:5] == r[:5] else None
                        if mine != r[:6] or strs != (r[6], r[7]):
                            ok = False
                    al.close()
                if not ok:
                    bad += 1
                    print("MISMATCH case %d vid %d job %d" % (case, vid, job), kw, env if knobs else "", flush=True)
        print("case %2d: genome %6d cov %4.1f err %.2f rep %.2f vols %d k=%d z=%d q=%d b=%d s=%d n=%d a=%d e=%.1f -> %d M4 (last vol) %s  %.1f s" % (
            case, genome, cov, err, rep, nv, kw["kmer_size"], kw["scan_window"], kw["kmer_cnt_cutoff"], kw["block_size"],
            kw["block_score_cutoff"], kw["num_candidates"], kw["align_size_cutoff"], kw["error"], nrec, "ok" if bad == bad0 else "BAD", time.time() - t0), flush=True)
print("fuzz_parity: %d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)
