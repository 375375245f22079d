cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "index" > $O/run37_tests.txt 2>&1; echo "tests rc $?"; tail -2 $O/run37_tests.txt
timeout 300 python bench.py --no-cpu-baseline --no-widened > $O/run37_bench.json 2> $O/run37_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/run37_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['phases_ms_per_step']['index'], d['roofline_index']['ms'] if 'roofline_index' in d else None)
PY
