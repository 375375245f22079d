cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
for i in 1 2; do
for v in nt base; do
  if [ $v = nt ]; then export NECAT_HIP_LIB=$PWD/gpurun_nt_libnecat_hip.so; else unset NECAT_HIP_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --no-widened > $O/ab_lib_${v}_$i.json 2> $O/ab_lib_${v}_$i.err; echo "$v $i rc $?"
done; done
python - <<'PY'
import json
for i in (1,2):
  for v in ('nt','base'):
    try:
        d=json.loads(open('gpurun_out/r03/ab_lib_%s_%d.json'%(v,i)).read().strip().splitlines()[-1])
        p=d['phases_ms_per_step']
        print(v, i, d['ms_per_step'], p['extend'], d['roofline'].get('avg_launch_ms'))
    except Exception as e: print(v, i, 'failed', e)
PY
