cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "NECAT_A_CHUNKS=1" "NECAT_A_CHUNKS=2" "NECAT_A_CHUNKS=3" "NECAT_A_CHUNKS=4" "NECAT_A_CHUNKS=6"; do echo "== $v"; env $v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['phases_ms_per_step'])"; done
NECAT_A_CHUNKS=4 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -3
