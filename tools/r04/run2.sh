cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_asm_plan.py -q -m gpu > $O/run2_plan.txt 2>&1; echo "plan tests rc $?"; tail -60 $O/run2_plan.txt
timeout 600 python -m pytest tests/test_gpu_asmpm.py tests/test_gpu_zz_asm_align.py -q -m gpu -x > $O/run2_asmpm.txt 2>&1; echo "asmpm tests rc $?"; tail -15 $O/run2_asmpm.txt
timeout 600 python - > $O/run2_asm_program.txt 2>&1 <<'PY'
import os, sys, time, subprocess, tempfile
sys.path.insert(0, os.getcwd())
from necat_amd import build, synth
rs = synth.simulate_reads(5_000_000, 20.0, seed=71, err=0.03, repeat_frac=0.05)
tmp = tempfile.mkdtemp()
wrk = os.path.join(tmp, "asm_vols")
synth.write_volume_dir(wrk, rs)
args = "-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400".split()
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([build.OC2ASMPM] + args + ["-t", "16", wrk, "0", os.path.join(tmp, "mine.m4")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, NECAT_TRACE="2", NECAT_CLI_TRACE="1"))
    print("rc", r.returncode, "wall %.2f s" % (time.time() - t0))
    print(r.stderr[-5000:])
print("records", sum(1 for _ in open(os.path.join(tmp, "mine.m4"), "rb")))
PY
tail -70 $O/run2_asm_program.txt
