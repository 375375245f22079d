cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alternative_kernel" > $O/run12_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/run12_tests.txt
timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu > $O/run12_full.txt 2>&1; echo "full rc $?"; tail -3 $O/run12_full.txt
timeout 300 python bench.py --no-cpu-baseline --no-widened > $O/run12_bench.json 2> $O/run12_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/run12_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['phases_ms_per_step'], d['roofline'].get('avg_launch_ms'), d['roofline']['frac'])
PY
