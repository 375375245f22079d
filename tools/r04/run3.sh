cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_asm_plan.py tests/test_gpu_asmpm.py -q -m gpu -x > $O/run3_tests.txt 2>&1; echo "tests rc $?"; tail -5 $O/run3_tests.txt
python - > $O/run3_gen.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from necat_amd import synth
rs = synth.simulate_reads(5_000_000, 20.0, seed=71, err=0.03, repeat_frac=0.05)
synth.write_volume_dir("/tmp/asm_vols", rs)
PY
A="-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400"
for rep in 1 2; do
  /usr/bin/time -f "wall %e s" env NECAT_TRACE=2 NECAT_CLI_TRACE=1 necat_amd/csrc/oc2asmpm $A -t 16 /tmp/asm_vols 0 /tmp/mine.m4 > $O/run3_prog$rep.out 2> $O/run3_prog$rep.err
  grep -v "asm plan " $O/run3_prog$rep.err | tail -12
done
grep "asm plan " $O/run3_prog2.err | tail -12
rm -rf $O/asm_kt; rocprofv3 --kernel-trace --stats -d $O/asm_kt -o r --output-format csv -- necat_amd/csrc/oc2asmpm $A -t 16 /tmp/asm_vols 0 /tmp/mine2.m4 > $O/asm_kt.log 2>&1
python tools/make_profiles.py stats $O/asm_kt $O/run3_asmpm_kernel_stats.md "rocprofv3 --kernel-trace --stats -- oc2asmpm $A -t 16 (5 Mb x 20, 3 % errors)"; rm -rf $O/asm_kt
head -40 $O/run3_asmpm_kernel_stats.md
