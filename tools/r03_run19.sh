cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alternative_kernel or pm_main" > $O/run19_tests.txt 2>&1; echo "tests rc $?"; tail -2 $O/run19_tests.txt
for v in 1; do NECAT_RC_FUSED=$v timeout 300 python bench.py --no-cpu-baseline --no-widened > $O/ab_fused_$v.json 2> $O/ab_fused_$v.err; echo "fused $v rc $?"; done
python - <<'PY'
import json
for v in (1,):
    d=json.loads(open('gpurun_out/r03/ab_fused_%d.json'%v).read().strip().splitlines()[-1])
    print(v, d['ms_per_step'], d['phases_ms_per_step'], d['roofline'].get('avg_launch_ms'))
PY
