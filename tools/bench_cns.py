#!/usr/bin/env python
"""Throughput of necat_cns_extension_batch on the bench workload: E. coli-size synthetic reads,
candidates from this library's own oc2pmov -j 0 path, role-swapped into one partition as oc2pcan does.

    python tools/bench_cns.py [genome_len coverage]
(the CPU port of the same loop is timed by tests/tools/cns_cpu_port.py)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from necat_amd import capi, synth  # noqa: E402


def main():
    args = sys.argv[1:]
    glen = int(args[0]) if len(args) > 0 else 4_600_000
    cov = float(args[1]) if len(args) > 1 else 40.0
    import util
    rs = synth.simulate_reads(glen, cov, seed=7)
    ctx = capi.Context(0)
    sizes = rs.sizes.astype(np.int64)
    off = np.zeros(sizes.shape[0], dtype=np.int64)
    off[1:] = np.cumsum(sizes)[:-1]
    vol = ctx.upload_volume(synth.pack_2bit(rs.codes), int(sizes.sum()), off, sizes)
    opt = capi.default_options(**dict(util.FAST, kmer_size=15, job=0))
    t = time.time()
    ix = ctx.build_index(vol, opt.kmer_size, opt.kmer_cnt_cutoff)
    c = ctx.find_candidates(ix, vol, vol, 0, 0, opt, True)
    ix.free()
    part = util.pcan_single_partition(capi.pack_candidates(c).tobytes())
    print("candidates: %d (%.2f s), partition records: %d" % (c.shape[0], time.time() - t, len(part) // 28), file=sys.stderr)
    t = time.time()
    cands, toff, n_all = ctx.cns_load_partition(vol, np.frombuffer(part, dtype=np.uint8))
    t_load = time.time() - t
    co = capi.cns_options()
    best = None
    for it in range(3):
        t = time.time()
        res = ctx.cns_extension_batch(vol, cands, toff, n_all, co)
        dt = time.time() - t
        ov = res.overlaps
        cols = int(ov["align_size"].sum())
        qb = int((ov["qend"] - ov["qoff"]).sum())
        line = dict(templates=int(res.templates.shape[0]), overlaps=int(ov.shape[0]), aligned=int(res.n_aligned), used=int(res.n_used),
                    rounds=int(res.n_rounds), wall_s=round(dt, 4), device_ms=round(res.device_ms, 1), host_ms=round(res.host_ms, 1),
                    accepted_columns=cols, accepted_query_bp=qb)
        res.free()
        if best is None or dt < best["wall_s"]:
            best = line
    best["load_partition_s"] = round(t_load, 3)
    best["templates_per_s"] = round(best["templates"] / best["wall_s"], 1)
    best["alignments_per_s"] = round(best["aligned"] / best["wall_s"], 1)
    import json
    print(json.dumps(best))


if __name__ == "__main__":
    main()
