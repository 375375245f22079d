"""Run one BASELINE.json-style configuration end to end on the GPU and print timings (tool, not a test)."""
import argparse, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from necat_amd import capi, synth
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--genome", type=int, default=12_000_000)
ap.add_argument("--coverage", type=float, default=50.0)
ap.add_argument("--scan-window", type=int, default=10)
ap.add_argument("--seed", type=int, default=13)
ap.add_argument("--job", type=int, default=1)
a = ap.parse_args()
t0 = time.time()
rs = synth.simulate_reads(a.genome, a.coverage, seed=a.seed)
print("generated %d reads / %d bp in %.1f s" % (rs.nreads, rs.nbases, time.time() - t0), flush=True)
ctx = capi.Context(0)
vol = ctx.upload_volume(synth.pack_2bit(rs.codes), rs.nbases, rs.offsets, rs.sizes)
opt = capi.default_options(**dict(bench.FAST, scan_window=a.scan_window, job=a.job))
for it in range(2):
    t0 = time.perf_counter()
    ix = ctx.build_index(vol, 15, 500); t1 = time.perf_counter()
    c = ctx.find_candidates(ix, vol, vol, 0, 0, opt, True); t2 = time.perf_counter()
    m = ctx.extend(vol, vol, 0, 0, c, opt, 1) if a.job == 1 else None; t3 = time.perf_counter()
    tm = ctx.timings(); ix.free()
    n = m.shape[0] if m is not None else c.shape[0]
    gbp = float((m["qend"] - m["qoff"]).sum()) / 1e9 if m is not None else 0.0
    print("iter %d: index %.1f ms seed %.1f ms extend %.1f ms (rounds %d) total %.1f ms | candidates %d records %d | %.0f overlaps/s %.3f Gbp/s" % (
        it, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), tm.rounds, 1e3*(t3-t0), c.shape[0], n, n/(t3-t0), gbp/(t3-t0)), flush=True)
