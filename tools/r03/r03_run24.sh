cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
for v in 16384 512 2048; do NECAT_RCWALK=$v timeout 300 python bench.py --no-cpu-baseline > $O/ab_rcthr3_$v.json 2> $O/ab_rcthr3_$v.err; done
python - <<'PY'
import json
for v in (16384, 512, 2048):
    try:
        d=json.loads(open('gpurun_out/r03/ab_rcthr3_%d.json'%v).read().strip().splitlines()[-1])
        p=d['phases_ms_per_step']
        print(v, d['ms_per_step'], p['extend'], d.get('oc2pmov_cold_start'), {k: (v2.get('wall_s') if isinstance(v2, dict) else v2) for k, v2 in d.get('widened_paths', {}).items()})
    except Exception as e: print(v, 'failed', e)
PY
