"""GPU parity of oc2asmpm: this repo's program - block vote and chained ranges on the device (necat_asm_plan_batch, asm_plan.h), every anchor of a
volume pair through the device's 2048-bp block aligner (necat_asm_align_batch: k_myers_ckg + k_rcwalk2w), DALIGNER's end extension on the host (rescue.h) - against
the REFERENCE's own oc2asmpm -t 1
(oracle/_ref/oc2asmpm, built from /root/reference; it travels to the GPU box): byte-identical text records, field-identical binary records."""
import os
import subprocess

import numpy as np
import pytest

from necat_amd import build, capi, synth
from oracle import oracle_api as ora

REF_ASMPM = os.path.join(os.path.dirname(ora.REF_PMOV), "oc2asmpm")
pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not os.path.exists(REF_ASMPM), reason="needs oracle/_ref/oc2asmpm")


def test_oc2asmpm_golden(built, tmp_path):
    """the committed vectors: tests/golden/asm_d = what the reference's oc2asmpm wrote for tests/golden/vols_d"""
    import json
    from tests import util
    built.build_cli()
    m = json.load(open(os.path.join(util.GOLDEN, "manifest_asm_rm.json")))["asm_d"]
    wrk = util.install_golden_volumes(m["volumes"], tmp_path)
    for v in range(m["n_volumes"]):
        got = os.path.join(str(tmp_path), "mine_%d.m4" % v)
        r = subprocess.run([build.OC2ASMPM] + m["args"].split() + ["-t", "3", wrk, str(v), got], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        assert open(got, "rb").read() == open(os.path.join(util.GOLDEN, "asm_d", "ref_v%d.m4" % v), "rb").read(), v



@needs_ref
@pytest.mark.parametrize("seed,err,repeat,indels,args", [
    (61, 0.03, 0.3, False, "-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400 -k 13"),      # necat.pl:36 (ASM_OVLP_OPTIONS)
    (62, 0.06, 0.4, True, "-z 5 -k 12 -n 20 -u 1"),                                     # TRIM_OVLP_OPTIONS write binary records (-u 1)
])
def test_oc2asmpm_reproduces_reference(built, tmp_path, seed, err, repeat, indels, args):
    built.build_cli()
    rs = synth.simulate_reads(40_000, 10.0, seed=seed, err=err, repeat_frac=repeat)
    if indels:
        rs = synth.add_long_indels(rs, 0.3, seed=seed + 1)
    wrk = os.path.join(str(tmp_path), "vols")
    nv = synth.write_volume_dir(wrk, rs, 200_000)
    assert nv >= 2
    total = 0
    for v in range(nv):
        want, got = os.path.join(str(tmp_path), "ref_%d.m4" % v), os.path.join(str(tmp_path), "mine_%d.m4" % v)
        subprocess.run([REF_ASMPM] + args.split() + ["-t", "1", wrk, str(v), want], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        env = dict(os.environ, NECAT_ASM_CALL_ANCHORS="70") if v == 0 and seed == 62 else os.environ      # once with many small calls of the block aligner
        r = subprocess.run([build.OC2ASMPM] + args.split() + ["-t", "4", wrk, str(v), got], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        assert r.returncode == 0, r.stderr
        a, b = open(want, "rb").read(), open(got, "rb").read()
        if "-u 1" in args:          # the reference leaves the records' 4 padding bytes uninitialised
            x, y = np.frombuffer(b, dtype=capi.M4_DTYPE), np.frombuffer(a, dtype=capi.M4_DTYPE)
            assert x.shape == y.shape
            for f in capi.M4_DTYPE.names:
                if not f.startswith("_"):
                    assert (x[f] == y[f]).all(), (v, f)
            total += x.shape[0]
        else:
            assert a == b, v
            total += len(a.splitlines())
    assert total > 300
