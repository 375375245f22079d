cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
for v in 2048 8192 16384; do NECAT_RC_POOL_MB=$v timeout 600 python bench.py --genome 12000000 --coverage 50 --steps 3 --warmup 1 --no-cpu-baseline --no-widened > $O/ab_pool_$v.json 2> $O/ab_pool_$v.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r03/ab_pool_$v.json').read().strip().splitlines()[-1])
p=d['phases_ms_per_step']
print($v, d['ms_per_step'], p['index'], p['seed'], p['extend'], d['config']['overlaps_per_step'])
PY
done
