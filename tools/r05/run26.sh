# round 5, call 26: defaults settled (several batches: two lanes at once; one batch: one lane; GPU_MAX_HW_QUEUES=8 set by the library): the whole GPU suite, smoke, the bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
s=$(date +%s)
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/run26_gpu_tests.txt 2>&1; echo "GPU suite rc $? in $(( $(date +%s) - s )) s"; tail -5 $O/run26_gpu_tests.txt | head -4
python -c "import __graft_entry__ as g; g.smoke()" > $O/run26_smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/run26_smoke.txt
timeout 1500 python bench.py > $O/run26_bench.json 2> $O/run26_bench.err; echo "bench rc $?"; tail -2 $O/run26_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05/run26_bench.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['value'], d['phases_ms_per_step'], d['roofline']['frac'], d['roofline_index']['frac'], d['roofline_seed']['frac'])
print(d.get('two_lanes_one_batch_cut'))
print(d['extra_configs']['configs2_sensitive'])
print(d.get('end_to_end_with_h2d'), d.get('oc2pmov_cold_start'))
PY
