# round 5, call 20: yeast size, both lanes at once: batch size, hardware queues
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export NECAT_EXT_OVERLAP_PCT=100
O=gpurun_out/r05; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout 900 python bench.py --genome 12000000 --coverage 50 --seed 11 --scan-window 10 --steps 3 --warmup 1 --no-cpu-baseline --no-widened --no-pmc > $O/run20_$n.json 2> $O/run20_$n.err; echo "bench $n ($*) rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run20_$n.json') if l.startswith('{"metric"')][-1])
print('   ', d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
}
run q4 NECAT_EXT_OVERLAP=1
run q8 GPU_MAX_HW_QUEUES=8
run q16 GPU_MAX_HW_QUEUES=16
run q8b393 GPU_MAX_HW_QUEUES=8 NECAT_BATCH=393216
run q8b1300 GPU_MAX_HW_QUEUES=8 NECAT_BATCH=1300000
run q8b500 GPU_MAX_HW_QUEUES=8 NECAT_BATCH=500000
run q8noorder GPU_MAX_HW_QUEUES=8 NECAT_EXT_ORDER=0
run q8m100 GPU_MAX_HW_QUEUES=8 NECAT_RC3_MIN=100000
