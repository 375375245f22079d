cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for lib in libnecat_hip.so libnecat_hip_w5.so libnecat_hip_w6.so libnecat_hip_w8.so; do echo "== $lib"
NECAT_HIP_LIB=$PWD/necat_amd/csrc/$lib timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['phases_ms_per_step'])"; done
