cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for v in "NECAT_CHAIN_WAVE=0" "NECAT_CHAIN_WAVE=1"; do echo "== $v"; env $v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['phases_ms_per_step'])"; done
