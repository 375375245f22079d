cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
python - > $O/run6_gen.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from necat_amd import synth
rs = synth.simulate_reads(5_000_000, 20.0, seed=71, err=0.03, repeat_frac=0.05)
synth.write_volume_dir("/tmp/asm_vols", rs)
PY
A="-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400"
rm -rf $O/asm_kt; rocprofv3 --kernel-trace --stats -d $O/asm_kt -o r --output-format csv -- necat_amd/csrc/oc2asmpm $A -t 16 /tmp/asm_vols 0 /tmp/mine2.m4 > $O/asm_kt.log 2>&1
python tools/make_profiles.py stats $O/asm_kt $O/run6_asmpm_kernel_stats.md "rocprofv3 --kernel-trace --stats -- oc2asmpm $A -t 16 (5 Mb x 20, 3 % errors)"; rm -rf $O/asm_kt
grep "k_asm\|k_seed_hits" $O/run6_asmpm_kernel_stats.md | cut -c1-60,300-
for rep in 1 2; do
  s=$(date +%s.%N)
  NECAT_CLI_TRACE=1 necat_amd/csrc/oc2asmpm $A -t 16 /tmp/asm_vols 0 /tmp/mine.m4 > $O/run6_prog$rep.out 2> $O/run6_prog$rep.err
  e=$(date +%s.%N); python3 -c "print(\"mine wall %.2f s\" % ($e - $s))"
  grep "oc2asmpm\]" $O/run6_prog$rep.err
done
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
for mode in concurrent serial; do
  if [ $mode = serial ]; then export NECAT_SERIAL=1; else unset NECAT_SERIAL; fi
  rm -rf $O/prof_$mode; rocprofv3 --kernel-trace --stats -d $O/prof_$mode -o r --output-format csv -- $CMD > $O/prof_$mode.log 2>&1
  python tools/make_profiles.py stats $O/prof_$mode $O/run6_kernel_stats_$mode.md "rocprofv3 --kernel-trace --stats -- $CMD ($mode streams)"
  python tools/make_profiles.py timeline $O/prof_$mode $O/run6_round_timeline_$mode.txt "one bench step kernel by kernel ($mode streams)"
  rm -rf $O/prof_$mode
  tail -1 $O/prof_$mode.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', d['ms_per_step'], d['config'].get('overlaps_per_step'), d['phases_ms_per_step'])"
done
unset NECAT_SERIAL
