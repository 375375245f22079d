cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python - <<'PY'
import os, sys, tempfile, subprocess, time
sys.path.insert(0, os.getcwd())
from necat_amd import synth, build
pmov, _ = build.build_cli()
tmp = tempfile.mkdtemp()
rs = synth.simulate_reads(4_600_000, 40.0, seed=1)
d = os.path.join(tmp, "v"); synth.write_volume_dir(d, rs)
argv = "-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -e 0.5 -t 8 -j 1 -u 0 -i 0".split()
for i in range(3):
    t0 = time.time()
    r = subprocess.run([pmov] + argv + [d, "0", os.path.join(tmp, "o")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, NECAT_CLI_TRACE="1"))
    print("run", i, round(time.time() - t0, 3)); print(r.stderr[-1500:])
PY
