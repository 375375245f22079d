cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python bench.py > $O/r04_bench_final.json 2> $O/r04_bench_final.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/r04_bench_final.json').read().strip().splitlines()[-1])
for k in ('value','gbp_aligned_per_s','ms_per_step','phases_ms_per_step'): print(k, d.get(k))
print(d['roofline']['frac'], d['roofline_index']['ms'], d['candidates_job0']['ms_per_step'], d['extra_configs']['configs2_sensitive']['m4_job1']['ms_per_step'])
print(d['widened_paths']['oc2asmpm']['wall_s'], d['widened_paths']['oc2asmpm']['speedup_program'], d['widened_paths']['oc2cns_program']['wall_s'], d['cpu_baseline']['value'])
PY
