cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_asm_plan.py tests/test_gpu_asmpm.py tests/test_gpu_shard.py -q -m gpu -x > $O/run5_tests.txt 2>&1; echo "tests rc $?"; tail -5 $O/run5_tests.txt
python - > $O/run5_gen.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from necat_amd import synth
rs = synth.simulate_reads(5_000_000, 20.0, seed=71, err=0.03, repeat_frac=0.05)
synth.write_volume_dir("/tmp/asm_vols", rs)
PY
A="-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400"
for rep in 1 2 3; do
  s=$(date +%s.%N)
  NECAT_TRACE=2 NECAT_CLI_TRACE=1 necat_amd/csrc/oc2asmpm $A -t 16 /tmp/asm_vols 0 /tmp/mine.m4 > $O/run5_prog$rep.out 2> $O/run5_prog$rep.err
  e=$(date +%s.%N); python3 -c "print(\"mine wall %.2f s\" % ($e - $s))"
  grep "oc2asmpm\]" $O/run5_prog$rep.err
done
grep "asm plan\|asm_align" $O/run5_prog3.err | tail -24
s=$(date +%s.%N); oracle/_ref/oc2asmpm $A -t 16 /tmp/asm_vols 0 /tmp/ref.m4 > /dev/null 2>&1; e=$(date +%s.%N); python3 -c "print(\"reference wall %.2f s\" % ($e - $s))"
sort /tmp/mine.m4 | md5sum; sort /tmp/ref.m4 | md5sum; wc -l /tmp/mine.m4 /tmp/ref.m4
