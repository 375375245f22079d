cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alternative_kernel or edlib_blocks" > $O/run17_tests.txt 2>&1; echo "tests rc $?"; tail -2 $O/run17_tests.txt
timeout 300 python bench.py --no-cpu-baseline --no-widened > $O/run17_bench.json 2> $O/run17_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/run17_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['phases_ms_per_step'], d['roofline'].get('avg_launch_ms'), d['roofline']['frac'])
PY
