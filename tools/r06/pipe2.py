# round 6: several volume pairs in flight on ONE device - D contexts, D host threads, each thread running whole steps (index build + seeding + extension, records on
# the host) on a context of its own.  One step's extension ends in ~ 15 rounds that are one block's dependent chain each on a mostly idle chip, and its index build /
# seeding are HBM- and latency-bound while the DP kernels are issue-bound: what does the chip do with D steps side by side?
#   python tools/r06/pipe2.py [steps per depth] [depths ...]
import os, sys, time, threading, hashlib
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from necat_amd import capi, synth

FAST = dict(kmer_size=15, scan_window=20, kmer_cnt_cutoff=500, block_size=2000, block_score_cutoff=3, num_candidates=500, align_size_cutoff=1000, error=0.5, use_hdr_as_id=0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
depths = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 1, 2]
job = int(os.environ.get("PIPE_JOB", "1"))
# (PIPE_GENOME / PIPE_COV / PIPE_Z: another workload, e.g. configs[2]'s 12 Mb x 50 at -z 10)
FAST["scan_window"] = int(os.environ.get("PIPE_Z", "20"))
rs = synth.simulate_reads(int(os.environ.get("PIPE_GENOME", "4600000")), float(os.environ.get("PIPE_COV", "40")), seed=int(os.environ.get("PIPE_SEED", "7")))
pac = synth.pack_2bit(rs.codes)
opt = capi.default_options(**dict(FAST, job=job, num_threads=1))
maxd = max(depths)
ctxs = [capi.Context(0) for _ in range(maxd)]
vols = [c.upload_volume(pac, rs.nbases, rs.offsets, rs.sizes) for c in ctxs]


def one(i):
    c, v = ctxs[i], vols[i]
    ix = c.build_index(v, opt.kmer_size, opt.kmer_cnt_cutoff)
    if job == 1:
        m4, _ = c.map_pair(ix, v, v, 0, 0, opt, True, 1)
    else:
        m4 = c.find_candidates(ix, v, v, 0, 0, opt, True)
    ix.free()
    return m4


def md5(m):
    if job == 1:
        return hashlib.md5(b"".join(sorted(capi.m4_text_lines(m)))).hexdigest()
    return hashlib.md5(m.tobytes()).hexdigest()


ref = None
for i in range(maxd):
    for _ in range(3):
        m = one(i)
    h = md5(m)
    ref = ref or h
    assert h == ref, "context %d gives other records" % i
print("records per step", m.shape[0], "md5", ref, flush=True)

for d in depths:
    cnt = [0] * d
    last = [None] * d
    todo = [steps * d]
    lock = threading.Lock()

    def work(i):
        while True:
            with lock:
                if todo[0] == 0:
                    return
                todo[0] -= 1
            last[i] = one(i)
            cnt[i] += 1
    th = [threading.Thread(target=work, args=(i,)) for i in range(d)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    n = sum(cnt)
    ok = all(md5(x) == ref for x in last if x is not None)
    print("depth %d: %d steps in %.3f s = %.2f ms per step (throughput), %.2f M records/s, per-thread steps %s, records identical %s" % (
        d, n, dt, 1e3 * dt / n, n * m.shape[0] / dt / 1e6, cnt, ok), flush=True)
