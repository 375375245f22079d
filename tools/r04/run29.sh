cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for cfg in "1 0" "1 1" "2 1" "3 1" "4 1" "2 0"; do
  set -- $cfg
  NECAT_RC_PIPE=$1 NECAT_RC_PRIO=$2 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/run29_bench_$1_$2.json 2> $O/run29_bench_$1_$2.err; echo "bench pipe=$1 prio=$2 rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r04/run29_bench_$1_$2.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'])
PY
done
