# round 3: the wave-per-block windowed walk (parity, threshold A/B on the bench workload, oc2asmpm timing)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cns.py tests/test_gpu_zz_asm_align.py tests/test_gpu_asmpm.py tests/test_gpu_rm.py -m gpu -q --timeout 600 2>&1 | tail -25 > $O/run4_tests.txt
tail -8 $O/run4_tests.txt
for T in 0 4096 12288 32768; do
  NECAT_WALK_WAVE=$T timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/ab_walkwave_$T.json 2> $O/ab_walkwave_$T.err
  python - <<PY
import json
d=json.loads(open('$O/ab_walkwave_$T.json').read().strip().splitlines()[-1])
print('WALK_WAVE=$T', d['ms_per_step'], d['config']['overlaps_per_step'], d['phases_ms_per_step'])
PY
done
for L in 0 0; do echo "== NECAT_ASM_LANE=$L"; NECAT_ASM_LANE=$L NECAT_TRACE=3 timeout 600 python tests/tools/bench_asmpm.py 400000 15 0.03 2>&1 | grep -v "^\[necat\] index\|seeding\|batch@" | tail -14; done > $O/r03_bench_asmpm.txt 2>&1
tail -16 $O/r03_bench_asmpm.txt
