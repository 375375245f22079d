cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
python - > $O/run7_gen.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from necat_amd import synth
rs = synth.simulate_reads(5_000_000, 20.0, seed=71, err=0.03, repeat_frac=0.05)
synth.write_volume_dir("/tmp/asm_vols", rs)
PY
A="-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400"
for budget in 67108864 16777216 67108864; do
  s=$(date +%s.%N)
  NECAT_ASM_VOTE_BUDGET=$budget NECAT_TRACE=2 NECAT_CLI_TRACE=1 necat_amd/csrc/oc2asmpm $A -t 16 /tmp/asm_vols 0 /tmp/mine.m4 > $O/run7_prog.out 2> $O/run7_prog_$budget.err
  e=$(date +%s.%N); python3 -c "print(\"budget $budget: mine wall %.2f s\" % ($e - $s))"
  grep "oc2asmpm\]\|arenas\|asm plan:" $O/run7_prog_$budget.err
done
grep "asm plan" $O/run7_prog_67108864.err | head -20
for lds in 0 4096 9216 15360 0; do
  NECAT_CK_LDS=$lds timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-widened > $O/run7_bench_lds$lds.json 2> $O/run7_bench_lds$lds.err
  python3 -c "
import json
d=json.loads(open('$O/run7_bench_lds$lds.json').read().strip().splitlines()[-1]); print('ck_lds', $lds, d['ms_per_step'], d['phases_ms_per_step'])"
done
