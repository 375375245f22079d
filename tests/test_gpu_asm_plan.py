"""GPU parity of oc2asmpm's candidate stage on the device (necat_asm_plan_batch, necat_amd/csrc/asm_plan.h: block vote, per-read order and cut, chained ranges)
against the host statement of the same steps (necat_amd/csrc/asm_core.h, which tests/test_asmpm.py pins to the reference's own oc2asmpm on the CPU), read by
read, through tests/host_core/check_asm_plan.cpp; the program as a whole is compared with the reference's in tests/test_gpu_asmpm.py."""
import os
import subprocess

import pytest

from necat_amd import build, synth
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def check_plan(built, tmp_path_factory):
    exe = os.path.join(str(tmp_path_factory.mktemp("asm_plan")), "check_asm_plan")
    subprocess.run([build._hipcc(), "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(util.ROOT, "tests", "host_core", "check_asm_plan.cpp"),
                    "-L" + build.CSRC, "-lnecat_hip", "-Wl,-rpath," + build.CSRC, "-lpthread"], check=True)
    return exe


@pytest.mark.parametrize("seed,genome,cov,err,repeat,indels,vol,args,env", [
    (71, 40_000, 10.0, 0.03, 0.3, False, 200_000, "-n 100 -z 10 -k 13", {}),                                   # necat.pl:36 (ASM_OVLP_OPTIONS)
    (72, 40_000, 10.0, 0.06, 0.4, True, 200_000, "-z 5 -k 12 -n 20", {}),                                      # TRIM_OVLP_OPTIONS, a tight cut
    (73, 300_000, 25.0, 0.01, 0.2, False, 4_000_000, "-n 100 -z 10 -k 13", {"NECAT_ASM_VOTE_BUDGET": "200000", "NECAT_ASM_SEED_BUDGET": "300000"}),   # several chunks / batches
    (74, 30_000, 30.0, 0.002, 0.6, False, 2_000_000, "-n 100 -z 10 -k 13", {}),                                # long exact matches, repeats: full blocks, many matches per pair
])
def test_plan_equals_host_statement(check_plan, tmp_path, seed, genome, cov, err, repeat, indels, vol, args, env):
    rs = synth.simulate_reads(genome, cov, seed=seed, err=err, repeat_frac=repeat)
    if indels:
        rs = synth.add_long_indels(rs, 0.3, seed=seed + 1)
    wrk = os.path.join(str(tmp_path), "vols")
    nv = synth.write_volume_dir(wrk, rs, vol)
    dump = os.path.join(str(tmp_path), "votes.bin")
    for v in range(nv):
        r = subprocess.run([check_plan] + args.split() + [wrk, str(v)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           env=dict(os.environ, NECAT_ASM_DUMP_VOTES=dump, **env))
        assert r.returncode == 0, r.stdout[-6000:]
        assert "0 reads differ" in r.stdout
