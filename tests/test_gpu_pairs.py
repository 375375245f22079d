"""Several volumes on several GPUs (both granularities; BASELINE configs[3] / [4] in the small):
  * necat_find_candidates_part / necat_map_pair_part: the shares of a pair add up to the pair, against the ORACLE;
  * bench.py --gpus 2 really runs two ranks (here on one device: gloo process group + HIP IPC data path) and reports the
    single-rank record count, in single-volume mode and in the pair-scheduled multi-volume mode - the latter's records
    against the oracle's pm_main over every volume;
  * oc2pm with two / three workers splits the volume pairs (pair_sched.h) and still writes the reference's records."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util
from oracle import oracle_api as ora

pytestmark = pytest.mark.gpu
BENCH = os.path.join(util.ROOT, "bench.py")


def _oracle_all_volumes(d, nv, kw, tmp, job):
    o = ora.options(**dict(kw, job=job, binary_output=1, num_threads=4))
    recs = []
    for vid in range(nv):
        p = os.path.join(str(tmp), "ora_%d_%d" % (job, vid))
        ora.pm_main(o, vid, d, p)
        recs += ora.sorted_records(p, ora.record_size(o))
    return sorted(recs)


def test_shares_of_a_pair_add_up_to_the_oracle_records(ctx, tmp_path):
    from necat_amd import capi, synth
    rs = synth.simulate_reads(160_000, 16.0, seed=11)
    d = os.path.join(str(tmp_path), "v")
    nv = synth.write_volume_dir_cuts(d, rs, synth.remainder_cuts(rs, 2))
    assert nv == 2
    _, _, vols = capi.load_volumes_info(d)
    kw = dict(util.FAST, kmer_size=13)
    ref = ctx.load_volume(vols[0][0]); qry = ctx.load_volume(vols[1][0])
    ix = ctx.build_index(ref, 13, kw["kmer_cnt_cutoff"])
    slots = 8
    for reads, rstart in ((ref, vols[0][1]), (qry, vols[1][1])):                # the self pair and a cross pair
        nreads = reads.nseq
        chunk = capi.pair_chunk_reads(nreads, slots)
        whole_c = ctx.find_candidates(ix, ref, reads, rstart, 0, capi.default_options(**dict(kw, job=0)), True)
        whole_m, _ = ctx.map_pair(ix, ref, reads, rstart, 0, capi.default_options(**dict(kw, job=1)), True, 1)
        parts_c, parts_m = [], []
        for lo, hi in ((0, 3), (3, 4), (4, 4), (4, 8)):                          # unequal shares, an empty one
            parts_c.append(ctx.find_candidates_part(ix, ref, reads, rstart, 0, capi.default_options(**dict(kw, job=0)), chunk, lo, hi, slots))
            parts_m.append(ctx.map_pair_part(ix, ref, reads, rstart, 0, capi.default_options(**dict(kw, job=1)), chunk, lo, hi, slots)[0])
        assert parts_c[2].shape[0] == 0 and parts_m[2].shape[0] == 0
        assert min(p.shape[0] for k, p in enumerate(parts_m) if k != 2) > 0
        key = lambda c: sorted(bytes(r) for r in capi.pack_candidates(c).astype("<u4"))
        assert key(np.concatenate(parts_c)) == key(whole_c) and whole_c.shape[0] > 200
        assert util.m4_key_rows(np.concatenate(parts_m)) == util.m4_key_rows(whole_m)
    ix.free(); ref.free(); qry.free()
    # the whole-pair calls themselves against the oracle (both pairs of reference volume 0 = the job of volume 0)
    c0, m0 = capi.pm_main(ctx, capi.default_options(**dict(kw, job=1)), 0, d)
    p = os.path.join(str(tmp_path), "o0")
    ora.pm_main(ora.options(**dict(kw, job=1, binary_output=1, num_threads=4)), 0, d, p)
    want = np.frombuffer(open(p, "rb").read(), dtype=capi.M4_DTYPE)
    assert util.m4_key_rows(m0) == util.m4_key_rows(want)


def _bench(args, env=None, timeout=900):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    r = subprocess.run([sys.executable, BENCH] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout carries exactly one JSON line:\n" + r.stdout[-2000:]
    return json.loads(lines[0])


SMALL = ["--genome", "300000", "--coverage", "16", "--kmer", "13", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-widened"]


def test_bench_gpus_2_runs_two_ranks_and_the_same_records(built):
    """`python bench.py --gpus 2` with no launcher around it: the script starts its own two ranks (NECAT_BENCH_ONE_DEVICE: both on
    device 0, so the records travel by HIP IPC) and reports the record count of the 1-rank run"""
    one = _bench(SMALL + ["--gpus", "1"])
    two = _bench(SMALL + ["--gpus", "2"], env={"NECAT_BENCH_ONE_DEVICE": "1", "NECAT_INDEX_SHARD": "1"})      # (the index in hash-range slices + all-gather, whatever the plan says for a volume this small)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["overlaps_per_step"] == one["config"]["overlaps_per_step"] > 500
    assert two["scaling"] == "strong" and two["multi_gpu"]["transport"] == "ipc"
    ranks = two["multi_gpu"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["query_reads"] > 0 and r["index_allgather_bytes"] > 0 for r in ranks)
    assert "roofline" in two and "roofline" in one
    assert two["multi_gpu"]["index_mode"] == "hash-range slices + all-gather"
    # the plan's own choice for a 4.8 Mbp volume on two ranks: every rank builds the whole table, nothing is exchanged - same records
    rep = _bench(SMALL + ["--gpus", "2"], env={"NECAT_BENCH_ONE_DEVICE": "1"})
    assert rep["config"]["overlaps_per_step"] == one["config"]["overlaps_per_step"]
    assert rep["multi_gpu"]["index_mode"].startswith("replicated") and all(r["index_allgather_bytes"] == 0 for r in rep["multi_gpu"]["ranks"])
    assert rep["multi_gpu"]["index_plan"]["replicate_ms"] < rep["multi_gpu"]["index_plan"]["shard_ms"]


def test_bench_steps_in_flight_report_the_same_records(built):
    """`bench.py --in-flight D` (the default is bench.IN_FLIGHT_DEFAULT at N = 1): the K timed steps are dealt to D contexts / host threads on one resident volume - every step returns the records
    of a step run alone, the line says how many were in flight and carries the one-at-a-time leg inside `roofline` / `config` (the objects the driver's record keeps whole)"""
    few = ["--genome", "300000", "--coverage", "16", "--kmer", "13", "--steps", "4", "--warmup", "0", "--no-cpu-baseline", "--no-widened"]
    one = _bench(few + ["--gpus", "1", "--in-flight", "1", "--no-pmc"])
    three = _bench(few + ["--gpus", "1", "--in-flight", "3", "--no-pmc"])
    assert one["config"]["steps_in_flight"] == 1 and "one_in_flight" not in one["roofline"]
    assert three["config"]["steps_in_flight"] == 3 and three["steps"] == 4
    assert three["config"]["overlaps_per_step"] == one["config"]["overlaps_per_step"] > 500
    leg = three["roofline"]["one_in_flight"]
    assert leg["ms_per_step"] > 0 and three["config"]["one_step_at_a_time"]["ms_per_step"] == leg["ms_per_step"]
    assert three["roofline"]["frac_one_in_flight"] == leg["frac"] and three["roofline"]["timed_region"]["steps_in_flight"] == 3
    assert three["candidates_job0"]["steps_in_flight"] == 3 and three["candidates_job0"]["records_per_step"] == one["candidates_job0"]["records_per_step"]


@pytest.mark.parametrize("job", [1, 0])
def test_bench_pairs_mode_records_equal_the_oracle(built, tmp_path, job):
    """--parallelism pairs on 1 and on 2 ranks: the union of the ranks' records = the oracle's records of all three volume jobs"""
    from necat_amd import capi, synth
    base = ["--genome", "240000", "--coverage", "18", "--kmer", "13", "--steps", "1", "--warmup", "0", "--parallelism", "pairs", "--volumes", "3", "--job", str(job)]
    rs = synth.simulate_reads(240_000, 18.0, seed=7)
    d = os.path.join(str(tmp_path), "v")
    assert synth.write_volume_dir_cuts(d, rs, synth.remainder_cuts(rs, 3)) == 3
    kw = dict(util.FAST, kmer_size=13)
    want = _oracle_all_volumes(d, 3, kw, tmp_path, job)
    for world in (1, 2):
        pre = os.path.join(str(tmp_path), "recs_w%d" % world)
        out = _bench(base + ["--gpus", str(world), "--dump-records", pre], env={"NECAT_BENCH_ONE_DEVICE": "1", "NECAT_INDEX_SHARD": "1"} if world > 1 else None)
        assert out["n_gpus"] == world and out["config"]["overlaps_per_step"] == len(want) > 500
        got = []
        per_rank = []
        for r in range(world):
            a = np.load("%s_%d.npy" % (pre, r))
            per_rank.append(a.shape[0])
            if job == 1:
                got += [bytes(x) for x in a.astype(capi.M4_DTYPE)]
            else:
                got += [bytes(x) for x in capi.pack_candidates(a).astype("<u4")]
        assert sorted(got) == want
        assert min(per_rank) > 0.2 * len(want) / world           # both ranks did a real share
        if world == 2:
            mg = out["multi_gpu"]
            assert [len(r["units"]) > 0 for r in mg["ranks"]] == [True, True]
            assert mg["teams"]["0"] == [0, 1]                     # volume 0's pairs span both ranks: its index was built sharded
            assert mg["ranks"][1]["index_allgather_bytes"] > 0


@pytest.mark.parametrize("gpus,sched", [("0,0", None), ("0,0,0", None), ("0", "pairs"), ("0,0", "volumes")])
@pytest.mark.parametrize("job", [1, 0])
def test_oc2pm_pair_schedule_writes_the_reference_records(tmp_path, built, gpus, sched, job):
    pmov, pm = built.build_cli()
    d = util.install_golden_volumes("vols_b", tmp_path)
    man = json.load(open(os.path.join(util.GOLDEN, "manifest.json")))
    o = ora.options(**dict(man["b_v0_m4_txt"]["options"], job=job, binary_output=1 - job))
    out = os.path.join(str(tmp_path), "all.out")
    env = dict(os.environ, NECAT_GPUS=gpus)
    if sched:
        env["NECAT_PM_SCHEDULE"] = sched
    r = subprocess.run([pm] + ora.opt_argv(o) + [d, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert r.returncode == 0, r.stderr
    if sched != "volumes":
        assert "unit(s) of job" in r.stdout
    want = []
    for vid in range(3):
        p = os.path.join(str(tmp_path), "o%d" % vid)
        ora.pm_main(o, vid, d, p)
        want += ora.sorted_records(p, ora.record_size(o)) if o.binary_output else open(p, "rb").read().splitlines(keepends=True)
    got = ora.sorted_records(out, ora.record_size(o)) if o.binary_output else open(out, "rb").read().splitlines(keepends=True)
    assert sorted(got) == sorted(want) and len(want) > 300
    # the m4 goldens of this dataset were written by the REFERENCE binary: the job of volume 0 is in there record for record
    if job == 1:
        gold = open(os.path.join(util.GOLDEN, man["b_v0_m4_txt"]["file"]), "rb").read().splitlines(keepends=True)
        assert set(gold) <= set(got)
    for vid in range(3):
        assert os.path.exists(os.path.join(d, "pm%d.finished" % vid))
    assert not [f for f in os.listdir(d) if f.startswith("pm_result_")]
