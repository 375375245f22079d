cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for cfg in "1 0" "2 9216" "3 9216" "2 19456" "4 9216" "1 9216"; do
  set -- $cfg
  NECAT_RC_PIPE=$1 NECAT_CK_LDS=$2 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/run35_bench_$1_$2.json 2> $O/run35_bench_$1_$2.err; echo "bench pipe=$1 ck_lds=$2 rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r04/run35_bench_$1_$2.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step']['extend'])
PY
done
