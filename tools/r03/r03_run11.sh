cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "alternative_kernel or pm_main or full" > $O/run11_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/run11_tests.txt
for v in 1 0; do NECAT_RC_CARRY=$v timeout 300 python bench.py --no-cpu-baseline --no-widened > $O/ab_carry_$v.json 2> $O/ab_carry_$v.err; echo "carry $v rc $?"; done
python - <<'PY'
import json
for v in (1,0):
    try:
        d=json.loads(open('gpurun_out/r03/ab_carry_%d.json'%v).read().strip().splitlines()[-1])
        print(v, d['ms_per_step'], d['phases_ms_per_step'], d['roofline'].get('avg_launch_ms'))
    except Exception as e: print(v, 'failed', e)
PY
timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu > $O/run11_full.txt 2>&1; echo "full rc $?"; tail -3 $O/run11_full.txt
