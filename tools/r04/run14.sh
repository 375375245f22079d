cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_asm_plan.py tests/test_gpu_asmpm.py tests/test_gpu_zz_asm_align.py -q -m gpu -x > $O/run14_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/run14_tests.txt
python - > $O/run14_gen.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from necat_amd import synth
rs = synth.simulate_reads(5_000_000, 20.0, seed=71, err=0.03, repeat_frac=0.05)
synth.write_volume_dir("/tmp/asm_vols", rs)
PY
A="-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400"
for b in 16777216 16777216 33554432 8388608 16777216 67108864 16777216; do
  s=$(date +%s.%N)
  NECAT_ASM_VOTE_BUDGET=$b NECAT_TRACE=2 NECAT_CLI_TRACE=1 necat_amd/csrc/oc2asmpm $A -t 16 /tmp/asm_vols 0 /tmp/mine.m4 > $O/run14_prog.out 2> $O/run14_prog_$b.err
  e=$(date +%s.%N); python3 -c "print(\"budget $b: mine wall %.2f s\" % ($e - $s))"
  grep "oc2asmpm\] pair\|arenas\|asm plan:" $O/run14_prog_$b.err | cut -c1-260
done
sort /tmp/mine.m4 | md5sum
