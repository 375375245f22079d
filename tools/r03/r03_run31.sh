cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
for v in 512 1024 2048 512 1024 2048; do NECAT_RCWALK=$v NECAT_TAIL_FUSED=$v timeout 300 python bench.py --no-cpu-baseline --no-widened > $O/ab_tail2_$v.json 2> $O/ab_tail2_$v.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r03/ab_tail2_$v.json').read().strip().splitlines()[-1])
p=d['phases_ms_per_step']
print($v, d['ms_per_step'], p['extend'], p['fused_tail_kernel'], p['fused_tail_launches'], p['fused_tail_blocks'])
PY
done
