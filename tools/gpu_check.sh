# GPU regression pass: the parity suite of the hot path + a short bench line (tool)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['phases_ms_per_step'])"
