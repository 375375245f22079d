cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_asm_plan.py tests/test_gpu_asmpm.py -q -m gpu -x > $O/run13_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/run13_tests.txt
python - > $O/run13_gen.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from necat_amd import synth
rs = synth.simulate_reads(5_000_000, 20.0, seed=71, err=0.03, repeat_frac=0.05)
synth.write_volume_dir("/tmp/asm_vols", rs)
PY
A="-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400"
for v in ov ov noov ov noov ov; do
  if [ $v = noov ]; then export NECAT_ASM_NO_OVERLAP=1; else unset NECAT_ASM_NO_OVERLAP; fi
  s=$(date +%s.%N)
  NECAT_TRACE=2 NECAT_CLI_TRACE=1 necat_amd/csrc/oc2asmpm $A -t 16 /tmp/asm_vols 0 /tmp/mine.m4 > $O/run13_prog.out 2> $O/run13_prog_$v.err
  e=$(date +%s.%N); python3 -c "print(\"$v: mine wall %.2f s\" % ($e - $s))"
  grep "oc2asmpm\]\|arenas\|asm plan:" $O/run13_prog_$v.err
done
unset NECAT_ASM_NO_OVERLAP
sort /tmp/mine.m4 | md5sum
