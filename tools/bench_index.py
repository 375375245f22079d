"""Time the index build alone on the E. coli-size volume (profiling tool)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from necat_amd import capi, synth
rs = synth.simulate_reads(4_600_000, 40.0, seed=7)
ctx = capi.Context(0)
vol = ctx.upload_volume(synth.pack_2bit(rs.codes), rs.nbases, rs.offsets, rs.sizes)
for it in range(4):
    ix = ctx.build_index(vol, 15, 500)
    ms = ctx.timings().index_ms
    ix.free()
print("index_ms %.2f" % ms)
