"""CPU test of the kernels' per-lane cores: necat_amd/csrc/{seed,dp,ext}_core.h are compiled with g++
(tests/host_core/check_core.cpp) and replayed lane by lane against the oracle - every candidate of
every read, every block alignment (distance, end column, traceback, tail trimming, identity)."""
import os
import subprocess

import pytest

from tests import util


@pytest.fixture(scope="module")
def check_core(tmp_path_factory, built):
    d = tmp_path_factory.mktemp("hc")
    exe = os.path.join(str(d), "check_core")
    obj = os.path.join(str(d), "necat_oracle.o")
    subprocess.run(["gcc", "-O2", "-std=gnu99", "-c", os.path.join(util.ROOT, "oracle", "necat_oracle.c"), "-o", obj], check=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                    os.path.join(util.ROOT, "tests", "host_core", "check_core.cpp"), obj, "-lm", "-lpthread"], check=True)
    return exe


@pytest.mark.parametrize("ds,vid,args", [
    ("vols_a", 0, "13 20 500 2000 3 500 1000 0.5".split()),
    ("vols_b", 0, "12 10 100 2000 3 500 1000 0.5".split()),
    ("vols_b", 1, "12 10 100 2000 3 3 400 0.5".split()),
    ("vols_a", 0, "11 5 500 1000 3 20 400 0.5 100000 0".split()),
])
def test_cores_match_oracle(check_core, tmp_path, ds, vid, args):
    d = util.install_golden_volumes(ds, tmp_path)
    r = subprocess.run([check_core, d, str(vid)] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "seed_mismatch=0 ext_mismatch=0 walk_mismatch=0" in r.stdout
    assert "candidates=0 " not in r.stdout


@pytest.mark.parametrize("err,args", [
    (0.03, "13 10 500 2000 3 100 400 0.5 150"),      # corrected reads, the options of necat.pl:36
    (0.13, "13 20 500 2000 3 40 400 0.5 60"),        # raw reads: wide bands at 2048 x 2048
])
def test_cores_at_block_2048_match_oracle(built, tmp_path, err, args):
    """the block aligner of oc2asmpm (asm_pm/blockwise_edlib.c = onc_align with 2048-bp blocks and tail match length 8)
    from the SAME per-lane cores: ext_plan<2048>, myers_block<32 / 44 words>, traceback_block / walk_block, ext_finish_block replayed
    on the CPU against the oracle's onc_align(2048, 8) - every candidate's coordinates, columns and identity"""
    d = str(tmp_path)
    obj = os.path.join(d, "necat_oracle.o")
    exe = os.path.join(d, "check_core2048")
    subprocess.run(["gcc", "-O2", "-std=gnu99", "-c", os.path.join(util.ROOT, "oracle", "necat_oracle.c"), "-o", obj], check=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-DCHECK_BLOCK=2048", "-DCHECK_TAIL=8", "-o", exe,
                    os.path.join(util.ROOT, "tests", "host_core", "check_core.cpp"), obj, "-lm", "-lpthread"], check=True)
    wrk, rs, nv = util.make_dataset(tmp_path, genome=60_000, coverage=12.0, seed=52, err=err, vol_size=400_000)
    r = subprocess.run([exe, wrk, "0"] + args.split(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "seed_mismatch=0 ext_mismatch=0 walk_mismatch=0" in r.stdout
    f = dict(kv.split("=") for kv in r.stdout.split()[1:])
    assert int(f["blocks"]) > 1000 and int(f["m4"]) > 300


@pytest.mark.parametrize("npairs,seed", [(60, 7), (45, 1234)])
def test_asm_kernel_source_on_the_cpu(built, tmp_path, npairs, seed):
    """asm_kernels.h (k_asm_align, the device's block aligner for oc2asmpm) compiled with g++ behind stand-ins for the HIP built-ins and run lane by
    lane, wave by wave, in launches of two waves that reuse the slabs - as necat_asm_align_batch launches it - against the oracle's onc_align at block
    size 2048 / tail 8: coordinates, identity and every column, anchors anywhere, both subject strands, unrelated sequences"""
    d = str(tmp_path)
    obj = os.path.join(d, "necat_oracle.o")
    exe = os.path.join(d, "check_asm_kernel")
    subprocess.run(["gcc", "-O2", "-std=gnu99", "-c", os.path.join(util.ROOT, "oracle", "necat_oracle.c"), "-o", obj], check=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(util.ROOT, "tests", "host_core", "check_asm_kernel.cpp"), obj,
                    "-lm", "-lpthread"], check=True)
    r = subprocess.run([exe, str(npairs), str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    f = dict(kv.split("=") for kv in r.stdout.split()[1:])
    assert int(f["mismatches"]) == 0 and int(f["aligned"]) > 100 and int(f["empty"]) > 3 and int(f["blocks"]) > 300


def test_asm_kernel_source_under_sanitizers(built, tmp_path):
    """the same harness built with -fsanitize=address,undefined: the slabs, the op pool and the column regions have exactly the sizes the library gives
    them, so an out-of-range access of the kernel is an error here"""
    d = str(tmp_path)
    obj = os.path.join(d, "necat_oracle.o")
    exe = os.path.join(d, "check_asm_kernel_asan")
    subprocess.run(["gcc", "-O2", "-std=gnu99", "-c", os.path.join(util.ROOT, "oracle", "necat_oracle.c"), "-o", obj], check=True)
    c = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", exe,
                        os.path.join(util.ROOT, "tests", "host_core", "check_asm_kernel.cpp"), obj, "-lm", "-lpthread"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if c.returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here")
    r = subprocess.run([exe, "30", "99"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stdout[-3000:]
    assert "mismatches=0" in r.stdout and "ERROR" not in r.stdout and "runtime error" not in r.stdout


def test_wave_chain_dp_model_equals_sequential(tmp_path):
    """chain_fill_wave (seed_kernels.h) turns the order-dependent predecessor scan of chain_dp.c:46-85 into prefix operations over
    64 lanes; its lane-by-lane host transcription must give the f / p / v of the sequential loop, max_skip stops included."""
    exe = os.path.join(str(tmp_path), "check_chain")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(util.ROOT, "include"), "-o", exe,
                    os.path.join(util.ROOT, "tests", "host_core", "check_chain.cpp")], check=True)
    r = subprocess.run([exe, "300"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert " mismatches 0 " in r.stdout


def test_record_writer_matches_printf(tmp_path):
    """host_fmt.h writes the M4 / candidate text lines without printf; its "%.2f" (exact integer rounding) and integer fields
    must be what printf gives."""
    src = os.path.join(str(tmp_path), "t.cpp")
    open(src, "w").write(r'''
#include "necat_amd/csrc/host_fmt.h"
#include <random>
int main() {
    std::mt19937_64 g(1); long bad = 0, n = 0; char a[512], b[512];
    auto chk = [&](double x) { *necat_host::put_f2(a, x) = 0; sprintf(b, "%.2f", x); ++n; if (strcmp(a, b)) { if (bad < 10) printf("%.17g: %s vs %s\n", x, a, b); ++bad; } };
    for (int i = 0; i < 3000000; ++i) chk((double)(g() % 10000001) / 100000.0);
    for (int i = 0; i < 300000; ++i) { double x = (g() % 100000) / 1000.0 + (g() % 8) * 0.125; chk(x); chk(x * 1e-3); chk(x * 1e-9); chk(100.0 * (double)(g() % 5000) / (double)(1 + g() % 5000)); }
    for (int i = 0; i <= 10000; ++i) { chk(i / 100.0); chk(i / 100.0 + 0.005); chk(i * 0.125); }
    chk(0.0); chk(1e-300); chk(99.995); chk(100.0); chk(4e15); chk(-1.5);
    for (int i = 0; i < 200000; ++i) {
        necat_m4 m; memset(&m, 0, sizeof m);
        m.qid = (int)(g() % 2000000) - 5; m.sid = (int)(g() % 2000000); m.ident_perc = (double)(g() % 100001) / 1000.0; m.vscore = (int)(g() % 100000);
        m.qdir = g() & 1; m.qoff = g() % 100000; m.qend = g() % 100000; m.qsize = g() % 1000000; m.sdir = 0; m.soff = g() >> (g() % 60); m.send = g() % 77; m.ssize = g() % 100000;
        *necat_host::put_m4(a, m, nullptr, nullptr) = 0;
        sprintf(b, "%d\t%d\t%.2f\t%d\t%d\t%lu\t%lu\t%lu\t%d\t%lu\t%lu\t%lu\n", m.qid, m.sid, m.ident_perc, m.vscore, m.qdir, m.qoff, m.qend, m.qsize, m.sdir, m.soff, m.send, m.ssize);
        ++n; if (strcmp(a, b)) { if (bad < 10) printf("%s vs %s", a, b); ++bad; }
    }
    printf("%ld checked, %ld bad\n", n, bad);
    return bad != 0;
}
''')
    exe = os.path.join(str(tmp_path), "t")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", util.ROOT, "-I", os.path.join(util.ROOT, "include"), "-o", exe, src, "-lpthread"], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout


# ---- the host side of necat_cns_extension_batch (necat_amd/csrc/cns_loop.h): select / replay vs the sequential loop ----

@pytest.fixture(scope="module")
def check_cns(tmp_path_factory, built):
    d = tmp_path_factory.mktemp("hcc")
    exe = os.path.join(str(d), "check_cns")
    objs = []
    for src in ("necat_oracle.c", "cns_oracle.c"):
        obj = os.path.join(str(d), src[:-2] + ".o")
        subprocess.run(["gcc", "-O2", "-std=gnu99", "-c", os.path.join(util.ROOT, "oracle", src), "-o", obj], check=True)
        objs.append(obj)
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                    os.path.join(util.ROOT, "tests", "host_core", "check_cns.cpp")] + objs + ["-lm", "-lpthread"], check=True)
    return exe


@pytest.mark.parametrize("args", [
    [],                                                  # defaults
    "400 4 12 0.5 0.8 1 3 6".split(),                    # fixed identity cutoff
    "400 2 6 0.5 0.8 0 -1 1".split(),                    # no speculation: nothing is aligned in vain
    "2000 4 12 0.5 0.5 0 40 50".split(),                 # speculate whole groups
    "400 4 30 0.5 0.8 0 1 2".split(),                    # deeper coverage than the data has
    "400 4 12 0.5 0.8 0 1 0".split(),                    # speculation width from the missing coverage
])
def test_cns_loop_matches_sequential(check_cns, tmp_path, args):
    """the batched loop (speculative selection + in-order replay) takes exactly the decisions of the sequential
    loop - overlaps, order, weights, gapped strings, cutoff, counters - whatever the speculation width"""
    import shutil
    wrk = util.install_golden_volumes("vols_c", tmp_path)
    for fn in ("cands.p0", "cands.partitions"):
        shutil.copy(os.path.join(util.GOLDEN, "cns_c", fn), os.path.join(str(tmp_path), fn))
    r = subprocess.run([check_cns, wrk, os.path.join(str(tmp_path), "cands")] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "cns_mismatch=0 templates=98 " in r.stdout
    assert " overlaps=0 " not in r.stdout
    if args[-2:] == ["-1", "1"]:
        f = dict(kv.split("=") for kv in r.stdout.split())
        assert f["aligned"] == f["used"]


def test_cns_loop_deep_coverage_cut_to_300(check_cns, tmp_path):
    """templates with more than MAX_EXAMINED_CAN = 300 candidates (tiny genome, 400x): the cut after the sort, the
    min_cov test on the count before the cut, several groups of 50 - batched loop vs sequential oracle"""
    from oracle import oracle_api as ora
    wrk, rs, nv = util.make_dataset(tmp_path, genome=5_000, coverage=400.0, seed=23, err=0.12)
    o = ora.options(**dict(util.FAST, job=0, binary_output=1, num_threads=4))
    rec = b""
    for v in range(nv):
        out = os.path.join(str(tmp_path), "pm_%d" % v)
        ora.pm_main(o, v, wrk, out)
        rec += open(out, "rb").read()
    part = util.pcan_single_partition(rec)
    import numpy as np
    per_template = np.bincount(np.frombuffer(part, dtype="<u4").reshape(-1, 7)[:, 1])
    assert per_template.max() > 300
    util.write_partition(os.path.join(str(tmp_path), "cands"), part)
    r = subprocess.run([check_cns, wrk, os.path.join(str(tmp_path), "cands")] + "400 4 30 0.5 0.8 0 1 12".split(),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "cns_mismatch=0 " in r.stdout and " overlaps=0 " not in r.stdout



# ---- the diagonal-band walk of k_rcwalk3 (necat_amd/csrc/ext_bandwalk.h): the per-lane cores against walk_block ----

@pytest.mark.parametrize("seed", [1, 20260928])
def test_band_walk_equals_walk_block(tmp_path, seed):
    """band_piece + band_walk_col - what a quad of k_rcwalk3 stores per column and what its walker does with it - replayed
    on the CPU over random blocks of every geometry (512 x 512, ragged, list B, 2048-bp blocks), error rates 0 - 35 %, long indels that force
    the redo path, every tail-match length, ops kept or not: the same n / nmat / tail statistics / ops as walk_block on the full matrix"""
    exe = os.path.join(str(tmp_path), "check_bandwalk")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(util.ROOT, "necat_amd", "csrc"), "-o", exe,
                    os.path.join(util.ROOT, "tests", "host_core", "check_bandwalk.cpp")], check=True)
    r = subprocess.run([exe, "1500", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "1500 blocks equal to walk_block" in r.stdout
