cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python - <<'PY'
import os, sys, subprocess, tempfile, time
sys.path.insert(0, os.getcwd())
from necat_amd import build, synth
build.build_cli()
tmp = tempfile.mkdtemp(prefix="asmpm_")
rs = synth.simulate_reads(5_000_000, 20.0, seed=71, err=0.03, repeat_frac=0.05)
wrk = os.path.join(tmp, "vols"); synth.write_volume_dir(wrk, rs)
args = "-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400".split()
for cache in (1, 0):
    for t in (8, 16, 32):
        env = dict(os.environ, NECAT_CLI_TRACE="1")
        if not cache: env["NECAT_ASM_NO_SUBJECT_CACHE"] = "1"
        t0 = time.time()
        r = subprocess.run([build.OC2ASMPM] + args + ["-t", str(t), wrk, "0", os.path.join(tmp, "o.m4")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        line = [l for l in r.stderr.splitlines() if l.startswith("[oc2asmpm]")]
        print("cache", cache, "threads", t, "wall %.2f" % (time.time() - t0), line[-1][10:] if line else r.stderr[-200:])
PY
