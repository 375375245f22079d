cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
tools/ck_microbench > $O/run25_ck_microbench.txt 2>&1; cat $O/run25_ck_microbench.txt
s=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/run25_parity.txt 2>&1; echo "parity rc $? in $(( $(date +%s) - s )) s"; tail -2 $O/run25_parity.txt
for post in 1 0 1; do
  NECAT_CK_POST=$post timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/run25_bench_$post.json 2> $O/run25_bench_$post.err; echo "bench post=$post rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r04/run25_bench_$post.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'])
PY
done
