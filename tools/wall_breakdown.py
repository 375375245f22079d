import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from necat_amd import capi, synth
import bench
opt = capi.default_options(**dict(bench.FAST, job=1, num_threads=1))
rs = synth.simulate_reads(4_600_000, 40.0, seed=7)
pac = synth.pack_2bit(rs.codes)
ctx = capi.Context(0)
vol = ctx.upload_volume(pac, rs.nbases, rs.offsets, rs.sizes)
for it in range(3):
    t0 = time.perf_counter(); ix = ctx.build_index(vol, 15, 500); t1 = time.perf_counter()
    ti = ctx.timings().index_ms
    c = ctx.find_candidates(ix, vol, vol, 0, 0, opt, True); t2 = time.perf_counter()
    ts = ctx.timings().seed_ms
    m = ctx.extend(vol, vol, 0, 0, c, opt, 1); t3 = time.perf_counter()
    te = ctx.timings().extend_ms
    ix.free(); t4 = time.perf_counter()
    print("index wall %.1f (ev %.1f) | seed wall %.1f (ev %.1f) | extend wall %.1f (ev %.1f) | free %.1f | total %.1f" % (
        1e3*(t1-t0), ti, 1e3*(t2-t1), ts, 1e3*(t3-t2), te, 1e3*(t4-t3), 1e3*(t4-t0)))
