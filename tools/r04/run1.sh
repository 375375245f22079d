cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "edlib_blocks or alternative_kernel" > $O/run1_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/run1_tests.txt
for ww in 1 0; do
  NECAT_RC_WW=$ww timeout 300 python bench.py --no-cpu-baseline --no-widened > $O/run1_bench_ww$ww.json 2> $O/run1_bench_ww$ww.err
done
python - <<'PY'
import json
for ww in (1, 0):
    try:
        d=json.loads(open('gpurun_out/r04/run1_bench_ww%d.json' % ww).read().strip().splitlines()[-1])
        print('ww', ww, d['ms_per_step'], d['phases_ms_per_step'], d.get('kernels_ms_per_step'))
    except Exception as e:
        print('ww', ww, 'failed', e)
PY
