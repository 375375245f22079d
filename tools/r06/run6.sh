# round 6, call 6: the two-lane scheduler with its result kernel left to an event (no host wait) and throttled polling - the several-batches cases and the multi-batch
# full-size md5s (yeast: 4 batches; the repeat-family set: 1.94 M candidates = 3 batches) three times over, then the whole GPU suite once
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
: > $O/run6_stress.txt
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -x -k "several_batches or yeast or ecoli_repeats" >> $O/run6_stress.txt 2>&1; echo "iteration $i rc $?"
done
grep "passed\|failed\|error" $O/run6_stress.txt | sort | uniq -c
s=$(date +%s)
timeout 2700 python -m pytest tests/ -q -m gpu -x > $O/run6_gpu_suite.txt 2>&1; echo "GPU suite rc $? in $(( $(date +%s) - s )) s"; tail -6 $O/run6_gpu_suite.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc > $O/run6_bench.json 2> $O/run6_bench.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06/run6_bench.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'])
print(d.get('extra_configs'))
print(d.get('candidates_job0'))
PY
