"""Micro-benchmark of the DP + traceback kernels alone: N synthetic 512 x 512 block alignments through
necat_edlib_align_batch (profiling tool, not part of the product path)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from necat_amd import capi
from necat_amd.synth import _mutate

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rng = np.random.default_rng(5)
base = rng.integers(0, 4, 4_000_000, dtype=np.uint8)
mut = _mutate(base, 0.13, rng)
# block i: target = base[o:o+512], query = a window of the mutated copy at the proportional place
n_src = 6000
offs = rng.integers(0, base.shape[0] - 1200, n_src)
seqs, qo, to = [], [], []
pos = 0
scale = mut.shape[0] / base.shape[0]
for o in offs:
    t = base[o:o + 512]
    qs = int(o * scale)
    # re-synchronise: mutate the target window directly (exact relation, 13 % errors)
    q = _mutate(base[o:o + 700], 0.13, rng)[:512]
    if q.shape[0] < 512:
        continue
    seqs += [q, t]
    qo.append(pos); pos += 512
    to.append(pos); pos += 512
seqs = np.concatenate(seqs)
reps = (N + len(qo) - 1) // len(qo)
qo = np.tile(np.asarray(qo, dtype=np.uint64), reps)[:N]
to = np.tile(np.asarray(to, dtype=np.uint64), reps)[:N]
ql = np.full(N, 512, dtype=np.int32)
ctx = capi.Context(0)
for it in range(3):
    t0 = time.perf_counter()
    dist, qe, te, _, _ = ctx.edlib_align_batch(seqs, qo, ql, to, ql, 0.5, want_ops=False)
    tm = ctx.timings()
    print("N=%d launches=%d myers %.3f ms traceback %.3f ms  word updates %.2f G  mean dist %.1f  fail %d  wall %.0f ms" % (
        N, tm.myers_launches, tm.myers_ms, tm.traceback_ms, tm.myers_word_updates / 1e9, dist[dist >= 0].mean(), int((dist < 0).sum()), 1e3 * (time.perf_counter() - t0)))
