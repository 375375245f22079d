cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for pr in 0 1 3 5 2 4 1; do
  NECAT_RC_PIPE=1 NECAT_RC_PRIO=$pr timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/run30_bench_$pr.json 2> $O/run30_bench_$pr.err; echo "bench prio=$pr rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r04/run30_bench_$pr.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'])
PY
done
