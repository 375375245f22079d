cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15 > $O/run9_tests.txt
tail -6 $O/run9_tests.txt
for T in 2048 8192 16384 49152; do
  NECAT_RCWALK=$T timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/ab_rcthr_$T.json 2> $O/ab_rcthr_$T.err
  python - <<PY
import json
d=json.loads(open('$O/ab_rcthr_$T.json').read().strip().splitlines()[-1])
print('RCWALK=$T', d['ms_per_step'], d['config']['overlaps_per_step'], d['phases_ms_per_step']['extend'])
PY
done
