"""round 5: why the cut-in-two step was 45.8 ms as bench.py's SECOND context (run21) and 37.1 ms as its only one (run19).
usage: ab_cut.py <order>   order = 'first' (the cut context is the process's first), 'second' (after a plain context ran two steps), 'second_closed' (.. and was closed)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from necat_amd import capi, synth
import bench
order = sys.argv[1] if len(sys.argv) > 1 else "second"
rs = synth.simulate_reads(4_600_000, 40.0, seed=7)
pac = synth.pack_2bit(rs.codes)
opt = capi.default_options(**dict(bench.FAST, job=1, num_threads=1))

def passes(ctx, n, tag):
    v = ctx.upload_volume(pac, rs.nbases, rs.offsets, rs.sizes)
    held = None
    for it in range(n):
        t0 = time.perf_counter()
        ix = ctx.build_index(v, 15, 500)
        m4, _ = ctx.map_pair(ix, v, v, 0, 0, opt, True, 1)
        ix.free()
        if held is None:
            held = m4
        print("%s pass %d: %.2f ms, extend %.2f ms, %d records" % (tag, it, 1e3 * (time.perf_counter() - t0), ctx.timings().extend_ms, m4.shape[0]), flush=True)
    v.free()

def cut_ctx():
    env = {"NECAT_EXT_OVERLAP_MIN": "131072", "NECAT_EXT_OVERLAP_SPLIT": "20", "NECAT_RC3_MIN": "130000"}
    os.environ.update(env)
    c = capi.Context(0)
    for k in env:
        os.environ.pop(k)
    return c

if order == "first":
    c2 = cut_ctx(); passes(c2, 6, "cut (first context)")
    c1 = capi.Context(0); passes(c1, 3, "plain (second context)")
else:
    c1 = capi.Context(0); passes(c1, 3, "plain (first context)")
    if order == "second_closed":
        c1.close()
    c2 = cut_ctx(); passes(c2, 6, "cut (second context)")
