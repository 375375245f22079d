cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/run22_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/run22_tests.txt
timeout 300 python bench.py --no-cpu-baseline --no-widened > $O/run22_bench.json 2> $O/run22_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/run22_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['phases_ms_per_step'], d['roofline'].get('avg_launch_ms'), d['roofline']['frac'])
PY
