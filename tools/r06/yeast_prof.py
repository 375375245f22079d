"""configs[2] size (12 Mb x 50, -z 10) alone: 1 warm-up + 3 passes of index -> map_pair, for rocprofv3 --kernel-trace --stats (which kernels carry the 283 ms)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from necat_amd import capi, synth
rs = synth.simulate_reads(12_000_000, 50.0, seed=11)
ctx = capi.Context(0)
vol = ctx.upload_volume(synth.pack_2bit(rs.codes), rs.nbases, rs.offsets, rs.sizes)
FAST = dict(kmer_size=15, scan_window=10, kmer_cnt_cutoff=500, block_size=2000, block_score_cutoff=3, num_candidates=500, align_size_cutoff=1000, ddfs_cutoff=0.25, error=0.5,
            num_output=500, use_hdr_as_id=0)
job = int(sys.argv[1]) if len(sys.argv) > 1 else 1
o = capi.default_options(**dict(FAST, job=job, num_threads=1))
for it in range(4):
    t0 = time.perf_counter()
    ix = ctx.build_index(vol, 15, 500)
    ti = ctx.timings().index_ms
    if job == 1:
        m4, nc = ctx.map_pair(ix, vol, vol, 0, 0, o, True, 1)
        n = m4.shape[0]
    else:
        n = ctx.find_candidates(ix, vol, vol, 0, 0, o, True).shape[0]
    tm = ctx.timings()
    ix.free()
    print("pass %d: %.1f ms wall, index %.1f seed %.1f extend %.1f, %d records" % (it, 1e3 * (time.perf_counter() - t0), ti, tm.seed_ms, tm.extend_ms, n), flush=True)
