"""oc2pmov -j 1 cold wall time against the band pool cap (tool; the VRAM a process dirties is paid for by the next one)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from necat_amd import build, synth
build.build_cli()
rs = synth.simulate_reads(4_600_000, 40.0, seed=7)
with tempfile.TemporaryDirectory() as td:
    d = os.path.join(td, "vols")
    synth.write_volume_dir(d, rs)
    out = os.path.join(td, "out")
    cmd = [build.OC2PMOV, "-k", "15", "-z", "20", "-q", "500", "-b", "2000", "-s", "3", "-n", "500", "-a", "1000", "-d", "0.25",
           "-e", "0.5", "-m", "500", "-t", "8", "-j", "1", "-u", "0", "-i", "0", d, "0", out]
    for mb in sys.argv[1:]:
        ts = []
        for it in range(4):
            t0 = time.perf_counter()
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, NECAT_BAND_POOL_MB=mb, NECAT_CLI_TRACE="1"))
            ts.append(time.perf_counter() - t0)
            last = r.stdout
        print("pool cap %5s MB: %s" % (mb, " ".join("%.3f" % t for t in ts)))
        print("   " + " | ".join(l.split("ms")[0].split("]")[1].strip() + " " + l.split("ms")[1].strip() for l in last.splitlines() if l.startswith("[pm]")))
