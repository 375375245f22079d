cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-widened --no-pmc > $O/run14_bench$i.json 2> $O/run14.err; echo "rc $?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run14_bench$i.json') if l.startswith('{"metric"')][-1])
print(d['steps'], d['ms_per_step'], d['value'], d['phases_ms_per_step']['index'], d['phases_ms_per_step']['seed'], d['phases_ms_per_step']['extend'])
PY
done
