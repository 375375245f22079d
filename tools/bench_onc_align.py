"""Throughput of necat_onc_align_batch (onc_align WITH alignment columns, tail_match_len = 4: the consensus
stage's call) on the bench workload's candidates; compared with necat_extend (coordinates only)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from necat_amd import capi, synth
import bench

genome = int(sys.argv[1]) if len(sys.argv) > 1 else 4_600_000
opt = capi.default_options(**dict(bench.FAST, job=1, num_threads=1))
rs = synth.simulate_reads(genome, 40.0, seed=7)
ctx = capi.Context(0)
vol = ctx.upload_volume(synth.pack_2bit(rs.codes), rs.nbases, rs.offsets, rs.sizes)
ix = ctx.build_index(vol, 15, 500)
cands = ctx.find_candidates(ix, vol, vol, 0, 0, opt, True)
print("candidates", cands.shape[0])
for it in range(3):
    t0 = time.perf_counter(); m4 = ctx.extend(vol, vol, 0, 0, cands, opt, 1); t1 = time.perf_counter()
    aln, ops, off = ctx.onc_align_batch(vol, vol, 0, 0, cands, opt, 4); t2 = time.perf_counter()
    gbp = float((aln["qend"] - aln["qoff"])[aln["ok"] == 1].sum()) / 1e9
    print("extend %.1f ms (%d M4) | onc_align_batch tail 4: %.1f ms, %d ok, %.2f Gbp aligned -> %.2f Gbp/s, %.2f G columns returned in %.2f GB" % (
        1e3 * (t1 - t0), m4.shape[0], 1e3 * (t2 - t1), int(aln["ok"].sum()), gbp, gbp / (t2 - t1), float(aln["align_size"].sum()) / 1e9, ops.shape[0] / 1e9))
