# round 6, first GPU call: (1) the any-order / in-order-dependency probe (can a kernel start while the previous one of its stream drains?),
# (2) the round's reference numbers on the untouched tree (bench line, no extras)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 120 tools/r06/anyorder_probe > $O/run1_probe.txt 2>&1; echo "probe rc $?"; cat $O/run1_probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-widened --no-pmc > $O/run1_bench.json 2> $O/run1_bench.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06/run1_bench.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['value'], d['phases_ms_per_step'])
PY
