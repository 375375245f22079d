"""Would two candidate cohorts in flight pay - the LONG chains (whose last rounds are latency bound and leave the chip empty)
extended next to the bulk?  Probe without code changes: two contexts on one device, two host threads, necat_extend of the two
subsets at the same time, against necat_extend of everything (tool)."""
import os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from necat_amd import capi, synth
import bench

rs = synth.simulate_reads(4_600_000, 40.0, seed=7)
c1, c2 = capi.Context(0), capi.Context(0)
pac = synth.pack_2bit(rs.codes)
v1 = c1.upload_volume(pac, rs.nbases, rs.offsets, rs.sizes)
v2 = c2.upload_volume(pac, rs.nbases, rs.offsets, rs.sizes)
opt = capi.default_options(**dict(bench.FAST, job=1))
ix = c1.build_index(v1, opt.kmer_size, opt.kmer_cnt_cutoff)
cands = c1.find_candidates(ix, v1, v1, 0, 0, capi.default_options(**dict(bench.FAST, job=0)), True)
ix.free()
right = np.minimum(cands["qsize"] - cands["qoff"], cands["ssize"] - cands["soff"]).astype(np.int64)
left = np.minimum(cands["qoff"], cands["soff"]).astype(np.int64)
blocks = right // 480 + left // 480
print("candidates %d, expected chain blocks: median %d, 90%% %d, max %d" % (cands.shape[0], np.median(blocks), np.percentile(blocks, 90), blocks.max()))

def timed(f, reps=4):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3

base = timed(lambda: c1.extend(v1, v1, 0, 0, cands, opt, 1))
print("one context, all candidates: %.1f ms" % base)
for T in (10, 14, 18, 22):
    long_ = np.ascontiguousarray(cands[blocks >= T]); rest = np.ascontiguousarray(cands[blocks < T])
    def both():
        th = threading.Thread(target=lambda: c2.extend(v2, v2, 0, 0, long_, opt, 1))
        th.start(); c1.extend(v1, v1, 0, 0, rest, opt, 1); th.join()
    both()
    t = timed(both)
    tl = timed(lambda: c2.extend(v2, v2, 0, 0, long_, opt, 1), 2); tr = timed(lambda: c1.extend(v1, v1, 0, 0, rest, opt, 1), 2)
    print("chains >= %2d blocks aside (%6d of %d): both at once %.1f ms | alone: long %.1f ms, rest %.1f ms" % (T, long_.shape[0], cands.shape[0], t, tl, tr))
