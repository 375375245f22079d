"""BASELINE configs[4]'s second half - "all-vs-all overlap FEEDING THE CNS CONSENSUS STAGE end-to-end" - as one chained test:

    reference chain   _ref/oc2pmov -j 0 -u 1 per volume  ->  cat  ->  _ref/oc2pcan  ->  _ref/oc2cns          (necat.pl:197-218, :240)
    this repo's chain oc2pm -j 0 -u 1 with the candidates partitioned ON THE DEVICE (NECAT_PM_PARTITIONS)  ->  oc2cns

on ONE workload of several volumes and several partitions; the corrected reads (cns_out) and the uncorrected rest (raw_out) must be the same
records (consensus_one_partition.c:110, consensus_aux.c:124, tasc/cbcns.c:47).  The reference writes its records in the order its threads finish
(with -t 1: template order, which is what this repo's oc2cns always writes), so the files are compared as sorted records.  k = 13 (a normal
flag): the reference's k = 15 table costs 66 s and 8.6 GB per volume whatever the input.
Needs oracle/_ref (the reference compiled from /root/reference; it travels to the GPU box, nothing here reads /root/reference)."""
import os
import subprocess
import time

import pytest

from necat_amd import build
from oracle import oracle_api as ora
from tests import util

pytestmark = pytest.mark.gpu


def _records(b: bytes):
    return sorted(b.split(b">"))


@pytest.mark.skipif(not (ora.have_ref() and ora.have_ref_cns() and os.path.exists(ora.REF_OC2CNS)),
                    reason="needs oracle/_ref (built from /root/reference; it travels to the GPU box)")
@pytest.mark.parametrize("genome,coverage,vol,part", [(2_000_000, 30.0, 20_000_000, 2500)])
def test_correction_chain_equals_the_reference_chain(built, tmp_path, genome, coverage, vol, part):
    built.build_cli()
    tmp = str(tmp_path)
    d, rs, nv = util.make_dataset(tmp_path, genome=genome, coverage=coverage, seed=61, err=0.12, vol_size=vol)
    assert nv >= 3
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    o = ora.options(**dict(util.FAST, kmer_size=13, job=0, binary_output=1, num_threads=cores))
    cns_argv = ora.cns_argv(ora.cns_options())
    # ---- the reference chain
    t0 = time.time()
    ref_can = os.path.join(tmp, "ref_cands")
    with open(ref_can, "wb") as f:
        for v in range(nv):
            ora.run_ref(o, v, d, ref_can + ".v%d" % v)
            f.write(open(ref_can + ".v%d" % v, "rb").read())
    ora.run_ref_pcan(d, ref_can, batch_size=part)
    nparts = int(open(ref_can + ".partitions").read().split()[0])
    assert nparts >= 3
    rc, rr = os.path.join(tmp, "ref_cns"), os.path.join(tmp, "ref_raw")
    subprocess.run([ora.REF_OC2CNS] + cns_argv + ["-t", str(cores), d, ref_can, rc, rr], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    t_ref = time.time() - t0
    # ---- this repo's chain: oc2pm writes the partitions itself (necat_pcan_partition on the resident candidates), oc2cns reads them
    t0 = time.time()
    our_can = os.path.join(tmp, "our_cands")
    r = subprocess.run([build.OC2PM] + ora.opt_argv(o) + [d, our_can], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, NECAT_PM_PARTITIONS=str(part)))
    assert r.returncode == 0, r.stderr
    assert int(open(our_can + ".partitions").read().split()[0]) == nparts
    oc, orw = os.path.join(tmp, "our_cns"), os.path.join(tmp, "our_raw")
    r = subprocess.run([build.OC2CNS] + cns_argv + ["-t", str(cores), d, our_can, oc, orw], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    t_our = time.time() - t0
    assert "beside the previous partition's consensus" in r.stdout          # the partitions went through the two-stage pipeline
    cns, raw = open(oc, "rb").read(), open(orw, "rb").read()
    ref_cns, ref_raw = open(rc, "rb").read(), open(rr, "rb").read()
    assert cns.count(b">") > 0.8 * rs.nreads                                # nearly every read is corrected at 30x
    assert _records(cns) == _records(ref_cns)
    assert _records(raw) == _records(ref_raw)
    # one partition after the other on the main thread: the same files
    r = subprocess.run([build.OC2CNS] + cns_argv + ["-t", str(cores), d, our_can, oc + "_seq", orw + "_seq"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, NECAT_CNS_PIPELINE="0"))
    assert r.returncode == 0, r.stderr
    assert open(oc + "_seq", "rb").read() == cns and open(orw + "_seq", "rb").read() == raw
    print("correction chain on %d reads / %d volumes / %d partitions: reference %.1f s, this repo %.1f s" % (rs.nreads, nv, nparts, t_ref, t_our))
