"""Stage-by-stage wall clock of the oc2pmov program on the bench workload, cold (tool; run on the GPU box)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from necat_amd import build, synth
build.build_cli()
rs = synth.simulate_reads(4_600_000, 40.0, seed=7)
with tempfile.TemporaryDirectory() as td:
    d = os.path.join(td, "vols")
    synth.write_volume_dir(d, rs)
    for job, binary in ((1, 0), (0, 1)):
        out = os.path.join(td, "out_%d" % job)
        cmd = [build.OC2PMOV, "-k", "15", "-z", "20", "-q", "500", "-b", "2000", "-s", "3", "-n", "500", "-a", "1000", "-d", "0.25",
               "-e", "0.5", "-m", "500", "-t", "1", "-j", str(job), "-u", str(binary), "-i", "0", d, "0", out]
        for it in range(3):
            t0 = time.perf_counter()
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, NECAT_CLI_TRACE="1", NECAT_TRACE=sys.argv[1] if len(sys.argv) > 1 else "0"))
            dt = time.perf_counter() - t0
            print("oc2pmov -j %d -u %d: %.3f s wall, rc %d, output %d bytes" % (job, binary, dt, r.returncode, os.path.getsize(out)))
            if it == 2:
                print("\n".join(l for l in r.stdout.splitlines() if l.startswith("[pm]") or l.startswith("[necat]"))[:6000])
