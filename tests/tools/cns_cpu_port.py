#!/usr/bin/env python
"""CPU baseline of the consensus stage's extension loop: the ORACLE's sequential restatement (one thread) on the first
templates of the partition tools/bench_cns.py measures on the GPU (same synthetic reads, same candidates - computed here
by the oracle's own oc2pmov -j 0 path, so this script needs no GPU).

    python tests/tools/cns_cpu_port.py [n_templates] [genome_len coverage]
"""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from necat_amd import capi, synth  # noqa: E402
from oracle import oracle_api as ora  # noqa: E402
import util  # noqa: E402


def main():
    k2 = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    glen = int(sys.argv[2]) if len(sys.argv) > 2 else 4_600_000
    cov = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
    k1 = max(1, k2 // 5)
    rs = synth.simulate_reads(glen, cov, seed=7)
    tmp = tempfile.mkdtemp(prefix="cns_cpu_")
    wrk = os.path.join(tmp, "vols")
    synth.write_volume_dir(wrk, rs, 1 << 40)
    o = ora.options(**dict(util.FAST, kmer_size=15, job=0, binary_output=1, num_threads=os.cpu_count() or 1))
    out = os.path.join(tmp, "pm_0")
    t = time.time()
    ora.pm_main(o, 0, wrk, out)
    rec = np.frombuffer(capi.pcan_single_partition(open(out, "rb").read()), dtype="<u4").reshape(-1, 7)
    print("candidates by the oracle: %d partition records (%.1f s)" % (rec.shape[0], time.time() - t), file=sys.stderr)
    ts = {}
    for k in (k1, k2):
        util.write_partition(os.path.join(tmp, "c%d" % k), rec[rec[:, 1] < k].tobytes())
        t = time.time()
        ora.cns_run(ora.cns_options(), wrk, os.path.join(tmp, "c%d" % k), os.path.join(tmp, "log%d" % k))
        ts[k] = time.time() - t
    n_al = sum(1 for ln in open(os.path.join(tmp, "log%d" % k2)) if ln[0] == "A") - sum(1 for ln in open(os.path.join(tmp, "log%d" % k1)) if ln[0] == "A")
    print(json.dumps({"cpu_port_templates_per_s_1thread": round((k2 - k1) / (ts[k2] - ts[k1]), 2),
                      "cpu_port_sample": "templates %d..%d of the partition, %d overlaps, %.1f s" % (k1, k2, n_al, ts[k2] - ts[k1])}))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
