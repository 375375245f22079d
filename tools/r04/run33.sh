cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for sp in 0 2 0 2; do
  NECAT_STREAM_PRIO=$sp timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/run33_bench_$sp.json 2> $O/run33_bench_$sp.err; echo "bench stream_prio=$sp rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r04/run33_bench_$sp.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'])
PY
done
