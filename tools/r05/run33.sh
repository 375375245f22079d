# round 5, call 33: stream priorities for the list-A / list-B chains (NECAT_STREAM_PRIO; round 4: nothing, with 4 hardware queues) now that every stream has a queue
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run33_$n.json 2> $O/run33_$n.err; echo "bench $n ($*) rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run33_$n.json') if l.startswith('{"metric"')][-1])
print('   ', d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
}
run base NECAT_STREAM_PRIO=0
run prioA NECAT_STREAM_PRIO=2
run prioB NECAT_STREAM_PRIO=1
run base2 NECAT_STREAM_PRIO=0
run prioA2 NECAT_STREAM_PRIO=2
run q16 GPU_MAX_HW_QUEUES=16
