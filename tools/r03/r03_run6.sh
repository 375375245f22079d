cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -15 > $O/run6_tests.txt
tail -6 $O/run6_tests.txt
for K in 1 0; do
  NECAT_SEED_KST=$K timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/ab_kst_$K.json 2> $O/ab_kst_$K.err
  python - <<PY
import json
d=json.loads(open('$O/ab_kst_$K.json').read().strip().splitlines()[-1])
print('SEED_KST=$K', d['ms_per_step'], d['config']['overlaps_per_step'], d['phases_ms_per_step'], d.get('roofline_index',{}).get('achieved'))
PY
done
