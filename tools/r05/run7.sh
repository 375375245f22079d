# round 5, call 7: the multi-rank paths with the index plan (replicate / slices), then the default bench line with its own PMC passes (traffic measured in the run)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_shard.py tests/test_gpu_pairs.py -q -x > $O/run7_shard.txt 2>&1; echo "shard + pairs rc $?"; tail -4 $O/run7_shard.txt
s=$(date +%s)
timeout 1500 python bench.py > $O/run7_bench.json 2> $O/run7_bench.err; echo "bench rc $? in $(( $(date +%s) - s )) s"; tail -3 $O/run7_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/run7_bench.json').read().strip().splitlines()[-1])
for k in ('value','gbp_aligned_per_s','ms_per_step','phases_ms_per_step'): print(k, d.get(k))
r=d['roofline']; print({k:r[k] for k in ('frac','achieved','traffic','avg_launch_ms','k_myers_ck','k_rcwalk')}); print(r['hbm'])
print(json.dumps(d.get('roofline_index'))[:1500])
print(json.dumps(d.get('roofline_seed'))[:2500])
print(d.get('candidates_job0'), d.get('oc2pmov_cold_start'))
PY
