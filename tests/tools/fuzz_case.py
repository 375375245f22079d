"""Re-run ONE case of tests/tools/fuzz_parity.py and show where the HIP path and the oracle differ (debugging tool).
    python tests/tools/fuzz_case.py <seed0> <case>"""
import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from necat_amd import capi, synth
from oracle import oracle_api as ora
from tests import util

seed0, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0 + case)
genome = int(rng.integers(40_000, 220_000)); cov = float(rng.uniform(8, 30)); err = float(rng.choice([0.04, 0.08, 0.12, 0.15]))
rep = float(rng.choice([0.0, 0.0, 0.05, 0.2])); multi = bool(rng.integers(0, 3) == 0)
kw = dict(kmer_size=int(rng.choice([11, 12, 13, 14])), scan_window=int(rng.choice([5, 10, 20])),
          kmer_cnt_cutoff=int(rng.choice([20, 100, 500])), block_size=int(rng.choice([1000, 2000, 3000])),
          block_score_cutoff=int(rng.choice([2, 3, 4])), num_candidates=int(rng.choice([3, 30, 500])),
          align_size_cutoff=int(rng.choice([500, 1000, 2000])), ddfs_cutoff=0.25, error=float(rng.choice([0.3, 0.5])),
          num_output=500, num_threads=2, use_hdr_as_id=0)
ctx = capi.Context(0)
with tempfile.TemporaryDirectory() as td:
    d, rs, nv = util.make_dataset(td, genome=genome, coverage=cov, seed=seed0 + case, err=err, repeat_frac=rep,
                                  vol_size=(max(300_000, int(genome * cov / 3)) if multi else synth.DEFAULT_VOL_SIZE))
    for vid in range(nv):
        o = ora.options(**dict(kw, job=0, binary_output=1))
        out = os.path.join(td, "o.out")
        ora.pm_main(o, vid, d, out)
        opt = capi.default_options(**dict(kw, job=0, binary_output=1))
        cands, _ = capi.pm_main(ctx, opt, vid, d)
        mine = sorted(bytes(r) for r in capi.pack_candidates(cands).astype("<u4"))
        ref = ora.sorted_records(out, 28)
        sm, sr = set(mine), set(ref)
        print("vid %d: %d mine, %d oracle, only mine %d, only oracle %d" % (vid, len(mine), len(ref), len(sm - sr), len(sr - sm)))
        def show(b):
            w = np.frombuffer(b, dtype="<u4")
            return "score %d flags %d | sid %d [%d,%d) | qid %d [%d,%d)" % (w[0] & ((1 << 29) - 1), w[0] >> 29, w[1], w[2], w[3], w[4], w[5], w[6])
        for b in sorted(sm - sr)[:6]: print("  mine  :", show(b))
        for b in sorted(sr - sm)[:6]: print("  oracle:", show(b))
