"""The (reference volume, query volume) pair scheduler (necat_amd/csrc/pair_sched.h through the C ABI's necat_pair_schedule:
host arithmetic, no device) and bench.py's launcher checks - CPU only."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util
from necat_amd import capi

SLOTS = 64
CASES = [
    ([184_000_000], 1), ([184_000_000], 4), ([184_000_000], 8),                 # one volume: its self pair split G ways
    ([2_000_000_000, 2_000_000_000, 1_600_000_000], 4),                        # Drosophila-like (BASELINE configs[3])
    ([1_050_000_000, 300_000_000, 130_000_000], 2), ([1_050_000_000, 300_000_000, 130_000_000], 3),
    ([2_000_000_000] * 44 + [1_300_000_000], 8),                               # human-like (configs[4]): 1035 pairs
    ([5, 7, 3, 9], 16),                                                        # more ranks than pairs
]


def _cost(vb, v, i):
    c = (vb[v] // 1024 + 1) * (vb[i] // 1024 + 1)
    return c * 0.5 if v == i else float(c)


@pytest.mark.parametrize("vb,G", CASES)
def test_schedule_covers_every_pair_once_and_balances(vb, G):
    units, off, team = capi.pair_schedule(vb, G, SLOTS)
    V = len(vb)
    assert off[0] == 0 and off[G] == len(units) and np.all(np.diff(off) >= 0)
    cover = {}
    load = np.zeros(G)
    for g in range(G):
        prev = None
        for u in units[off[g]:off[g + 1]]:
            v, i, lo, hi = int(u["ref_vol"]), int(u["query_vol"]), int(u["slot_lo"]), int(u["slot_hi"])
            assert 0 <= v <= i < V and 0 <= lo < hi <= SLOTS
            if prev is not None:
                assert (v, i) > prev, "a rank's units follow the job order"
            prev = (v, i)
            cover.setdefault((v, i), []).append((lo, hi, g))
            load[g] += _cost(vb, v, i) * (hi - lo) / SLOTS
            assert team[v, 0] <= g <= team[v, 1]
    # every pair: its slot ranges tile [0, SLOTS) exactly, on consecutive ranks
    assert sorted(cover) == [(v, i) for v in range(V) for i in range(v, V)]
    for key, parts in cover.items():
        parts.sort()
        assert parts[0][0] == 0 and parts[-1][1] == SLOTS
        for a, b in zip(parts, parts[1:]):
            assert a[1] == b[0] and b[2] > a[2]
    # balance: nobody carries more than an equal share plus one slot of the heaviest pair it touches
    total = sum(_cost(vb, v, i) for v in range(V) for i in range(v, V))
    slot_max = max(_cost(vb, v, i) for v in range(V) for i in range(v, V)) / SLOTS
    assert abs(load.sum() - total) < 1e-6 * total
    assert load.max() <= total / G + 1.01 * slot_max
    # a reference volume's team = the consecutive ranks that hold units of it
    for v in range(V):
        ranks = sorted({g for g in range(G) for u in units[off[g]:off[g + 1]] if int(u["ref_vol"]) == v})
        assert ranks == list(range(int(team[v, 0]), int(team[v, 1]) + 1))


def test_drosophila_shape_uses_all_four_gpus():
    """three volumes on four GPUs: whole reference volumes would leave one GPU idle and finish 3 : 2 : 1"""
    units, off, team = capi.pair_schedule([2_000_000_000, 2_000_000_000, 1_600_000_000], 4, SLOTS)
    assert all(off[g + 1] > off[g] for g in range(4))
    assert team[0].tolist() == [0, 2]            # volume 0's three pairs span three ranks: a sharded index build of three


def test_chunk_rule():
    assert capi.pair_chunk_reads(250_000) == 64 and capi.pair_chunk_reads(7_700) == 15 and capi.pair_chunk_reads(100) == 1


def test_bench_refuses_a_world_that_is_not_gpus():
    """`--gpus 8` under a launcher that started one rank must not report one rank's work as N = 8"""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert r.returncode != 0 and "--gpus 8" in r.stderr and r.stdout.strip() == ""


def test_index_plan_replicates_small_volumes_and_shards_big_ones_on_many_ranks(built, monkeypatch):
    """necat_index_plan (include/necat_hip.h), the cost model behind necat_index_build_sharded's choice: one rank builds an E. coli-size table in
    5 ms while the all-gather of its 3.3 GB takes longer for every N <= 4 - there every rank builds the whole table; a volume at oc2mkdb's 2 Gbp
    cut on 8 ranks is cheaper in hash-range slices.  Host arithmetic only: runs without a device."""
    from necat_amd import capi
    monkeypatch.delenv("NECAT_INDEX_SHARD", raising=False)
    monkeypatch.delenv("NECAT_XGMI_GBS", raising=False)
    ecoli, big = 184_010_740, 2_000_000_000
    p1 = capi.index_plan(ecoli, 15, 1)
    assert p1.shard == 0 and abs(p1.replicate_ms - 5.2) < 0.3 and p1.exchange_ms == 0.0
    for n in (2, 4):
        p = capi.index_plan(ecoli, 15, n)
        assert p.shard == 0 and p.shard_ms > p.replicate_ms and 3.0e9 < p.exchange_bytes < 3.6e9
    assert capi.index_plan(big, 15, 8).shard == 1 and capi.index_plan(big, 15, 2).shard == 0
    assert abs(capi.index_plan(big, 15, 1).replicate_ms - 58.0) < 3.0          # measured: 58 ms (profiles/r05_config4_human_subset.json)
    # a faster link moves the break-even; the environment overrides the choice
    assert capi.index_plan(ecoli, 15, 8, 400.0).shard == 1 and capi.index_plan(ecoli, 15, 8, 50.0).shard == 0
    monkeypatch.setenv("NECAT_INDEX_SHARD", "1")
    assert capi.index_plan(ecoli, 15, 2).shard == 1 and capi.index_plan(ecoli, 15, 1).shard == 0
    monkeypatch.setenv("NECAT_INDEX_SHARD", "0")
    assert capi.index_plan(big, 15, 8).shard == 0


@pytest.mark.parametrize("tsan", [False, True])
def test_pair_lanes_of_a_job_with_a_fake_library(tmp_path, tsan):
    """tests/host_core/check_pm_lanes.cpp: the SOURCE of necat_amd/csrc/pm_job.h (a job's units on NECAT_PAIR_LANES lanes, one context and host thread per lane, the
    records written in unit order) compiled with g++ against a fake C ABI whose mappings sleep - longest for the first units - and log their concurrency: same bytes at
    1 / 2 / 3 / 5 lanes and both jobs, min(L, units) mappings really side by side with lane l on units l, l + L, ..., everything freed, a failing unit or a lane without
    a context -> exit 1, no file, nobody left waiting.  tsan: the same under ThreadSanitizer (no report).  What the lanes do on the device is
    tests/test_gpu_cli_golden.py::test_pair_lanes_write_the_same_file."""
    import subprocess
    from necat_amd import synth
    rs = synth.simulate_reads(60_000, 12.0, seed=3)
    d = os.path.join(str(tmp_path), "vols")
    assert synth.write_volume_dir_cuts(d, rs, [rs.nbases // 5] * 4) == 5
    exe = os.path.join(str(tmp_path), "check_pm_lanes")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_core", "check_pm_lanes.cpp")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17"] + (["-fsanitize=thread"] if tsan else []) + ["-I" + os.path.join(util.ROOT, "include"), "-o", exe, src, "-lpthread"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if tsan and r.returncode != 0 and "tsan" in r.stdout.lower():
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    assert r.returncode == 0, r.stdout[-3000:]
    r = subprocess.run([exe, d, os.path.join(str(tmp_path), "out")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.strip().endswith("ok")
    assert "ThreadSanitizer" not in r.stderr
    assert r.stderr.count("ERROR") == 2 * (3 + 1)          # the injected failures, reported once each: three lane modes with a failing unit + one lane without a context, per job
