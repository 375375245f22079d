cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cns.py -x -q > $O/pytest_g.log 2>&1; echo "pytest rc $?"; tail -25 $O/pytest_g.log | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/bench_g.json 2> $O/bench_g.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02/bench_g.json'))
for k in ('value','ms_per_step','phases_ms_per_step'): print(k, d.get(k))
PY
