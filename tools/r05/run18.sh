# round 5, call 18: two lanes at yeast size (BASELINE configs[2]: 0.6 Gbp, -z 10: 2.46 M candidates = four batches of 615 k, until now one after the other)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout 900 python bench.py --genome 12000000 --coverage 50 --seed 11 --scan-window 10 --steps 3 --warmup 1 --no-cpu-baseline --no-widened --no-pmc > $O/run18_$n.json 2> $O/run18_$n.err; echo "bench $n ($*) rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run18_$n.json') if l.startswith('{"metric"')][-1])
print('   ', d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
}
run ov0 NECAT_EXT_OVERLAP=0
run ov0q8 NECAT_EXT_OVERLAP=0 GPU_MAX_HW_QUEUES=8
run p70 GPU_MAX_HW_QUEUES=8
run p70q4 NECAT_EXT_OVERLAP=1
run p50 GPU_MAX_HW_QUEUES=8 NECAT_EXT_OVERLAP_PCT=50
run p30 GPU_MAX_HW_QUEUES=8 NECAT_EXT_OVERLAP_PCT=30
run p90 GPU_MAX_HW_QUEUES=8 NECAT_EXT_OVERLAP_PCT=90
run p100 GPU_MAX_HW_QUEUES=8 NECAT_EXT_OVERLAP_PCT=100
run p70b393 GPU_MAX_HW_QUEUES=8 NECAT_BATCH=393216
run ov0b1300 NECAT_EXT_OVERLAP=0 NECAT_BATCH=1300000
