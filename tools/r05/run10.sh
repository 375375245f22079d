# round 5, call 10: the whole GPU suite after the knob refactor (per-context knobs), the 512-thread index splits, the arena guard; smoke; default bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
s=$(date +%s)
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/run10_gpu_tests.txt 2>&1; echo "GPU suite rc $? in $(( $(date +%s) - s )) s"; tail -6 $O/run10_gpu_tests.txt | head -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/run10_smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/run10_smoke.txt
timeout 1500 python bench.py > $O/run10_bench.json 2> $O/run10_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/run10_bench.json').read().strip().splitlines()[-1])
for k in ('value','gbp_aligned_per_s','ms_per_step','phases_ms_per_step'): print(k, d.get(k))
r=d['roofline']; print({k:r[k] for k in ('frac','achieved','traffic','avg_launch_ms','k_myers_ck','k_rcwalk')})
print(d['roofline_index']['frac'], d['roofline_index']['ms'], d['roofline_seed']['frac'], d['roofline_seed']['lookups_per_s'])
print(d.get('candidates_job0'), d.get('oc2pmov_cold_start'), d.get('end_to_end_with_h2d'))
print({k: d['cpu_baseline'].get(k) for k in ('value','cores','cpu_quota_cores','mapping_s','t1')})
PY
