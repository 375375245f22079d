"""round 6: the cold start of oc2pmov on a FRESH box (0.748 s on the driver's box against 0.343 s on the builder's).  This must be the
FIRST GPU work of the gpurun call: run 0 is the process nobody warmed anything for (no page cache for the runtime's libraries and code objects, GPU idle)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from necat_amd import build, synth
rs = synth.simulate_reads(4_600_000, 40.0, seed=7)
d = tempfile.mkdtemp(prefix="cold_")
vd = os.path.join(d, "vols")
synth.write_volume_dir(vd, rs)
pmov = build.OC2PMOV
base = ["-k", "15", "-z", "20", "-q", "500", "-b", "2000", "-s", "3", "-n", "500", "-a", "1000", "-d", "0.25", "-e", "0.5", "-m", "500", "-t", "1"]
for job, binary in ((1, 0), (0, 1), (1, 0)):
    for rep in range(4):
        env = dict(os.environ, NECAT_CLI_TRACE="1")
        t0 = time.time()
        r = subprocess.run([pmov] + base + ["-j", str(job), "-u", str(binary), "-i", "0", vd, "0", os.path.join(d, "out")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        print("== -j %d -u %d run %d: %.3f s rc %d" % (job, binary, rep, time.time() - t0, r.returncode))
        print("\n".join(l for l in r.stderr.splitlines() if "cli" in l.lower() or "ms" in l)[-2500:])
