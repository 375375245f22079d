cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
s=$(date +%s)
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/r04_final_gpu_tests.txt 2>&1; echo "GPU suite rc $? in $(( $(date +%s) - s )) s"; tail -6 $O/r04_final_gpu_tests.txt | head -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/r04_smoke.txt
timeout 1500 python bench.py > $O/r04_bench_final.json 2> $O/r04_bench_final.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/r04_bench_final.json').read().strip().splitlines()[-1])
for k in ('value','gbp_aligned_per_s','ms_per_step','phases_ms_per_step'): print(k, d.get(k))
print(d['roofline']['frac'], d['roofline']['traffic'])
print(d['widened_paths'].get('oc2asmpm'))
print(d.get('extra_configs'))
print(d.get('candidates_job0'), d.get('oc2pmov_cold_start'))
PY
