cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
for v in 16384 8192 4096 2048 1024 512; do NECAT_RCWALK=$v timeout 300 python bench.py --no-cpu-baseline --no-widened > $O/ab_rcthr2_$v.json 2> $O/ab_rcthr2_$v.err; done
python - <<'PY'
import json
for v in (16384, 8192, 4096, 2048, 1024, 512):
    try:
        d=json.loads(open('gpurun_out/r03/ab_rcthr2_%d.json'%v).read().strip().splitlines()[-1])
        p=d['phases_ms_per_step']
        print(v, d['ms_per_step'], p['extend'], p['myers_kernel'], p['traceback_kernel'], p['rcwalk_kernel'])
    except Exception as e: print(v, 'failed', e)
PY
