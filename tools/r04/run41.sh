cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for pr in 1 8 24 0 8 1; do
  NECAT_RC_PRIO=$pr timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/run41_bench_$pr.json 2> $O/run41_bench_$pr.err; echo "bench prio=$pr rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r04/run41_bench_$pr.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step']['extend'], d['phases_ms_per_step']['rcwalk_kernel'])
PY
done
