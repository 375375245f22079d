cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for pf in 0 1 0 1; do
  NECAT_RC_PREFETCH=$pf timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/run40_bench_$pf.json 2> $O/run40_bench_$pf.err; echo "bench prefetch=$pf rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r04/run40_bench_$pf.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step']['extend'], d['phases_ms_per_step']['rcwalk_kernel'])
PY
done
