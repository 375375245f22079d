cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "alternative or m4_matches or onc_align" 2>&1 | tail -15 > $O/run7_tests.txt
tail -12 $O/run7_tests.txt
for T in 0 16384; do
  NECAT_RCWALK=$T timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/ab_rc_$T.json 2> $O/ab_rc_$T.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/ab_rc_$T.json').read().strip().splitlines()[-1])
    print('RCWALK=$T', d['ms_per_step'], d['config']['overlaps_per_step'], d['phases_ms_per_step'])
except Exception as e:
    print('RCWALK=$T failed', e, open('$O/ab_rc_$T.err').read()[-600:])
PY
done
