cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
tail -3 $O/bench_default.time
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/bench_default.json').read().strip().splitlines()[-1])
for k in ('value','gbp_aligned_per_s','ms_per_step','phases_ms_per_step','candidates_job0','oc2pmov_cold_start'): print(k, d.get(k))
r=d['roofline']; print({k:r[k] for k in r if k not in ('note','issue_bound_note')})
print(d.get('roofline_index'))
print(d.get('widened_paths'))
print(d.get('cpu_baseline'))
PY
