"""round 5: where the 0.32 / 0.40 s of a cold oc2pmov run go.  Writes the bench volume, runs the program with its stage clocks on."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from necat_amd import build, synth
rs = synth.simulate_reads(4_600_000, 40.0, seed=7)
d = tempfile.mkdtemp(prefix="cold_")
vd = os.path.join(d, "vols")
synth.write_volume_dir(vd, rs)
pmov, _ = build.build_cli()
base = ["-k", "15", "-z", "20", "-q", "500", "-b", "2000", "-s", "3", "-n", "500", "-a", "1000", "-d", "0.25", "-e", "0.5", "-m", "500", "-t", "1"]
for job, binary in ((1, 0), (0, 1)):
    for rep in range(3):
        env = dict(os.environ, NECAT_CLI_TRACE="1", NECAT_TRACE="2" if rep == 2 else "0")
        t0 = time.time()
        r = subprocess.run([pmov] + base + ["-j", str(job), "-u", str(binary), "-i", "0", vd, "0", os.path.join(d, "out")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        print("== -j %d -u %d run %d: %.3f s rc %d" % (job, binary, rep, time.time() - t0, r.returncode))
        if rep >= 1:
            print(r.stderr[-6000:])
