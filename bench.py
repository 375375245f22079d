#!/usr/bin/env python
"""bench.py - overlaps/sec of the all-vs-all overlap stage (oc2pmov path) on MI355X.

One "step" = one full pass of the hot path over one reference volume that is already resident in
HBM: k-mer index build -> candidate search (both strands of every read) -> block-wise banded Myers
extension -> M4 records back on the host.  Workload at N=1 = BASELINE.json configs[1]:
E. coli-size (4.6 Mb) 40x synthetic ONT reads, OVLP_FAST_OPTIONS with -j 1 (M4 output).

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns one independent
reference volume of the same size (seed + rank) - the unit necat.pl itself distributes
(necat.pl:190-202) - so there is no data-path collective and scaling is weak; the barrier and the
max-over-ranks time follow the driver's contract.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_LANE_OPS_PER_S = 256 * 4 * 32 * 2.4e9   # 256 CUs x 4 SIMD-32 x 2.4 GHz (32-bit integer lane-ops)
OPS_PER_WORD_UPDATE = 45       # 32-bit VALU ops of one 64-row Myers word update

FAST = dict(kmer_size=15, scan_window=20, kmer_cnt_cutoff=500, block_size=2000, block_score_cutoff=3,
            num_candidates=500, align_size_cutoff=1000, ddfs_cutoff=0.25, error=0.5, num_output=500,
            use_hdr_as_id=0)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome", type=int, default=4_600_000)
    ap.add_argument("--coverage", type=float, default=40.0)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--kmer", type=int, default=15)
    ap.add_argument("--scan-window", type=int, default=20)
    ap.add_argument("--job", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-widened", action="store_true", help="skip the extra measurements of the SURVEY 8f.1 rows")
    ap.add_argument("--cpu-genome", type=int, default=2_300_000, help="genome size of the bounded CPU-baseline sample")
    return ap.parse_args()


def dist_setup(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run (also with one rank)
        import torch
        import torch.distributed as dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist_.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        dist = dist_
    return rank, world, local, dist


def barrier_sync(dist, local):
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize(local)
    except ImportError:
        pass
    if dist is not None:
        dist.barrier()


def cpu_baseline(args, opt_kw):
    """The reference's own oc2pmov (oracle/_ref, built from /root/reference) - or, when absent, the
    oracle port - timed on this host's cores on a bounded sample of the same workload."""
    from necat_amd import synth
    from oracle import oracle_api as ora
    tmp = tempfile.mkdtemp(prefix="necat_cpu_")
    rs = synth.simulate_reads(args.cpu_genome, args.coverage, seed=args.seed)
    # the reference hands out reads in chunks of 500 (pm_worker.c:13,354): more threads than chunks stay idle
    cores = max(1, min(os.cpu_count() or 1, (rs.nreads + 499) // 500))
    d = os.path.join(tmp, "vols")
    synth.write_volume_dir(d, rs)
    o = ora.options(**dict(opt_kw, job=args.job, binary_output=0, num_threads=cores))
    out = os.path.join(tmp, "out.txt")
    kind = "reference" if ora.have_ref() else "port"
    t0 = time.time()
    if kind == "reference":
        t_map = ora.run_ref(o, 0, d, out)
        wall = time.time() - t0
    else:
        st = ora.pm_main(o, 0, d, out)
        wall = time.time() - t0
        t_map = st.t_map
    nrec = sum(1 for _ in open(out, "rb"))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return {"value": round(nrec / max(t_map, 1e-9), 1), "unit": "overlaps/s", "cores": cores, "kind": kind,
            "sample": "%.2f Mb genome x %.0fx (%d reads, %d bp), same options, -t %d (= number of 500-read chunks, of %d host "
                      "threads); mapping phase %.2f s (index build excluded, as in the reference's own 'pairwise mapping' "
                      "timer); whole process %.1f s" % (args.cpu_genome / 1e6, args.coverage, rs.nreads, rs.nbases, cores,
                                                        os.cpu_count() or 1, t_map, wall),
            "overlaps": nrec, "mapping_s": round(t_map, 3), "whole_process_s": round(wall, 2)}


def widened_paths(ctx, vol, capi, opt_kw):
    """necat_onc_align_batch (onc_align with its gapped strings) and necat_cns_extension_batch (the consensus stage's
    extension loop for all templates of a partition) on this volume's own candidates"""
    opt0 = capi.default_options(**dict(opt_kw, job=0, num_threads=1))
    ix = ctx.build_index(vol, opt0.kmer_size, opt0.kmer_cnt_cutoff)
    cands = ctx.find_candidates(ix, vol, vol, 0, 0, opt0, True)
    ix.free()
    res = {}
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        aln, ops, off = ctx.onc_align_batch(vol, vol, 0, 0, cands, opt0, 4)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    gbp = float((aln["qend"] - aln["qoff"])[aln["ok"] == 1].sum()) / 1e9
    res["onc_align_batch"] = {"alignments": int(cands.shape[0]), "ms": round(1e3 * best, 2), "gbp_aligned_per_s": round(gbp / best, 3),
                              "columns": int(aln["align_size"].sum()), "column_bytes": int(ops.shape[0])}
    del aln, ops, off
    part = capi.pcan_single_partition(capi.pack_candidates(cands).tobytes())
    pc, toff, n_all = ctx.cns_load_partition(vol, np.frombuffer(part, dtype=np.uint8))
    co = capi.cns_options()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        r = ctx.cns_extension_batch(vol, pc, toff, n_all, co)
        dt = time.perf_counter() - t0
        line = {"templates": int(r.templates.shape[0]), "overlaps_accepted": int(r.overlaps.shape[0]), "alignments_computed": int(r.n_aligned),
                "alignments_sequential_loop": int(r.n_used), "passes": int(r.n_rounds), "ms": round(1e3 * dt, 2),
                "device_ms": round(r.device_ms, 2), "host_ms": round(r.host_ms, 2)}
        r.free()
        if best is None or dt < best[0]:
            best = (dt, line)
    best[1]["templates_per_s"] = round(best[1]["templates"] / best[0], 1)
    res["cns_extension_loop"] = best[1]
    return res


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON result: whatever libraries print there (RCCL announces its path on
    # stdout when the first communicator is created) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank, world, local, dist = dist_setup(args)
    from necat_amd import build, capi, synth
    if rank == 0:
        build.build_hip()          # no-op when the in-tree library is current; one rank only, the others wait
    if dist is not None:
        dist.barrier()
    opt_kw = dict(FAST, kmer_size=args.kmer, scan_window=args.scan_window)
    opt = capi.default_options(**dict(opt_kw, job=args.job, num_threads=1))
    # ---- synthetic volume of this rank, made resident in HBM before the clock starts
    rs = synth.simulate_reads(args.genome, args.coverage, seed=args.seed + 1000 * rank)
    pac = synth.pack_2bit(rs.codes)
    ctx = capi.Context(local)
    vol = ctx.upload_volume(pac, rs.nbases, rs.offsets, rs.sizes)

    def step():
        ix = ctx.build_index(vol, opt.kmer_size, opt.kmer_cnt_cutoff)
        t_index = ctx.timings().index_ms
        if args.job == 1:      # pm_search_one_volume of a mapping job: seeding + extension in one call, candidates stay on the device
            m4, _ = ctx.map_pair(ix, vol, vol, 0, 0, opt, True, 1)
            cands = None
        else:
            cands, m4 = ctx.find_candidates(ix, vol, vol, 0, 0, opt, True), None
        tm = ctx.timings()
        ix.free()
        return cands, m4, t_index, tm

    # setup (untimed): initialise the torch/HIP runtimes and let the library size its HBM pools once
    # (the traceback band pool alone is tens of GB: allocating + zeroing it is a one-off of ~2 s)
    barrier_sync(dist, local)
    step()
    for _ in range(args.warmup):
        step()
    barrier_sync(dist, local)
    t0 = time.perf_counter()
    agg = dict(index_ms=0.0, seed_ms=0.0, extend_ms=0.0, myers_ms=0.0, traceback_ms=0.0, launches=0, blocks=0, words=0, bases=0, rounds=0,
               a_ms=0.0, a_launches=0, a_blocks=0, tb_a_ms=0.0, big_ms=0.0, big_blocks=0)
    n_over = 0
    gbp = 0.0
    for _ in range(args.steps):
        cands, m4, t_index, tm = step()
        n_over += (m4.shape[0] if m4 is not None else cands.shape[0])
        if m4 is not None:
            gbp += float((m4["qend"] - m4["qoff"]).sum()) / 1e9
        agg["index_ms"] += t_index; agg["seed_ms"] += tm.seed_ms; agg["extend_ms"] += tm.extend_ms
        agg["myers_ms"] += tm.myers_ms; agg["traceback_ms"] += tm.traceback_ms; agg["launches"] += tm.myers_launches
        agg["blocks"] += tm.myers_blocks; agg["words"] += tm.myers_word_updates; agg["bases"] += tm.myers_cells_bases
        agg["rounds"] += tm.rounds
        agg["tb_a_ms"] += tm.tracebackA_ms
        if tm.myersA_big_blocks >= agg["big_blocks"]:
            agg["big_blocks"], agg["big_ms"] = int(tm.myersA_big_blocks), float(tm.myersA_big_ms)
        agg["a_ms"] += tm.myersA_ms; agg["a_launches"] += tm.myersA_launches; agg["a_blocks"] += tm.myersA_blocks
    barrier_sync(dist, local)
    elapsed = time.perf_counter() - t0
    from necat_amd import shard
    elapsed, tot_over, tot_gbp = shard.reduce_step_stats(dist, elapsed, float(n_over), gbp, device="cuda" if dist is not None else None)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    K = max(1, args.steps)
    # ---- roofline of the dominant kernel: k_myers_coop<8,16,512,8>, the DP of the full 512 x 512 blocks
    # (list A).  It and its traceback k_traceback<8,16,512,..> / the list-B pair are the four largest
    # entries of the rocprof summary (profiles/r01_kernel_stats.md), the DP kernel being the one that
    # does the arithmetic of the path.  Algorithmic HBM bytes of one block alignment = its two 2-bit
    # fragments in (2 x 512 / 4 B) + one 16-byte result out (SURVEY.md 8d "extension" row restated per
    # block); everything else the kernel moves is the traceback band it stores (`traffic`, from the PMC
    # passes kept in profiles/).
    A_BYTES_PER_BLOCK = 2 * 512 / 4.0 + 16.0
    a_launches = max(1, agg["a_launches"])
    avg_launch_ms = agg["a_ms"] / a_launches
    alg_per_launch = A_BYTES_PER_BLOCK * agg["a_blocks"] / a_launches
    achieved = alg_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    word_rate = agg["words"] / (agg["myers_ms"] * 1e-3) if agg["myers_ms"] > 0 else 0.0
    traffic = None
    tb_traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))

        def pmc_bytes(prefix):
            k = next(v for n, v in pmc.items() if n.startswith(prefix))
            # FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section)
            return (2.0 * k["FETCH_SIZE_KB_per_launch"] + k["WRITE_SIZE_KB_per_launch"]) * 1024.0
        traffic = pmc_bytes("void necat::k_myers_coop<8, 16, 512, 8, false>")
        tb_traffic = pmc_bytes("void necat::k_traceback<8, 16, 512, 1024, false>")
    except Exception:
        pass
    tb_avg_ms = agg["tb_a_ms"] / a_launches
    roofline = {"bound": "hbm", "kernel": "k_myers_coop<8,16,512,8,false>", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "traffic_frac": round(traffic / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and avg_launch_ms > 0 else None,
                "launches": int(agg["a_launches"]), "avg_launch_ms": round(avg_launch_ms, 4),
                "algorithmic_bytes_per_launch": round(alg_per_launch, 1),
                "traceback_kernel": {"kernel": "k_traceback<8,16,512,1024,false>", "avg_launch_ms": round(tb_avg_ms, 4), "traffic": tb_traffic,
                                     "traffic_frac": round(tb_traffic / (tb_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tb_traffic and tb_avg_ms > 0 else None},
                # the launch with the most blocks (throughput regime; the small launches of the late rounds are latency bound):
                # ~2 x 512 columns x 8 words per block (list A also holds shorter last blocks, so this slightly overstates)
                "biggest_launch": {"blocks": agg["big_blocks"], "ms": round(agg["big_ms"], 4),
                                   "valu_frac": round(agg["big_blocks"] * 8192.0 * OPS_PER_WORD_UPDATE / (agg["big_ms"] * 1e-3) / VALU_LANE_OPS_PER_S, 4) if agg["big_ms"] > 0 else None},
                "all_dp_kernels": {"launches": int(agg["launches"]), "ms": round(agg["myers_ms"], 2), "blocks": int(agg["blocks"]),
                                   "word_updates_per_s": round(word_rate, 1),
                                   "valu_frac": round(word_rate * OPS_PER_WORD_UPDATE / VALU_LANE_OPS_PER_S, 4)},
                "note": "integer DP: the algorithmic HBM fraction is small by construction (SURVEY.md 8d); the measured traffic (PMC, "
                        "profiles/r01_pmc_hbm_traffic.json) is the stored traceback band (16-byte records, only words that can lie on an "
                        "alignment of <= the block's distance) - traffic_frac is that traffic over the launch time against the HBM peak; "
                        "valu_frac = word updates x %d lane-ops / (256 CU x 128 lanes x 2.4 GHz)" % OPS_PER_WORD_UPDATE}
    out = {
        "metric": "overlaps/sec (all-vs-all, index build + seeding + banded Myers extension -> M4)",
        "value": round(tot_over / elapsed, 1), "unit": "overlaps/s",
        "gbp_aligned_per_s": round(tot_gbp / elapsed, 4),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / K, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "E. coli-size %.1f Mb genome, %.0fx synthetic ONT reads (12%% errors), %d reads / %d bp per GPU, "
                               "OVLP_FAST_OPTIONS (-k %d -z %d -q 500 -b 2000 -s 3 -n 500 -a 1000 -e 0.5) with -j %d; one reference volume per GPU"
                               % (args.genome / 1e6, args.coverage, rs.nreads, rs.nbases, args.kmer, args.scan_window, args.job),
                   "overlaps_per_step_per_gpu": n_over // K, "parallelism": "volume-per-gpu x%d" % world},
        "phases_ms_per_step": {"index": round(agg["index_ms"] / K, 2), "seed": round(agg["seed_ms"] / K, 2),
                               "extend": round(agg["extend_ms"] / K, 2), "myers_kernel": round(agg["myers_ms"] / K, 2),
                               "traceback_kernel": round(agg["traceback_ms"] / K, 2), "rounds": agg["rounds"] // K},
        "device": ctx.device_name(),
        "roofline": roofline,
    }
    if world == 1 and not args.no_widened:
        # SURVEY 8f.1 rows built on the same kernels, measured right AFTER the timed region (before the CPU baseline, while the GPU clocks are still up) on the same resident volume; reported
        # extras, not part of `value`
        try:
            out["widened_paths"] = widened_paths(ctx, vol, capi, opt_kw)
        except Exception as e:
            out["widened_paths"] = {"error": str(e)}
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args, opt_kw)
        except Exception as e:  # the baseline is a reported extra; never fail the GPU measurement on it
            out["cpu_baseline"] = {"value": None, "unit": "overlaps/s", "cores": os.cpu_count(), "kind": "unavailable", "sample": str(e)}
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
