# round 3, last check of the committed tree: GPU suite, smoke, default bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r03_final_gpu_tests.txt 2>&1; echo "tests rc $?"; grep -n "passed\|failed" $O/r03_final_gpu_tests.txt | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r03_final_smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/r03_final_smoke.txt
timeout 900 python bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/r03_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['phases_ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline'].get('value'))
PY
