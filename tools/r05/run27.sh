# round 5, call 27: the walk of piece i beside the pass of piece i + 1 (NECAT_RC_PIPE, round 4: slower) once more now that the streams have hardware queues of their own
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run27_$n.json 2> $O/run27_$n.err; echo "bench $n ($*) rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run27_$n.json') if l.startswith('{"metric"')][-1])
print('   ', d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
}
run base NECAT_RC_PIPE=1
run pipe2 NECAT_RC_PIPE=2
run pipe3 NECAT_RC_PIPE=3
run pipe4 NECAT_RC_PIPE=4
run pipe2q4 NECAT_RC_PIPE=2 GPU_MAX_HW_QUEUES=4
run pipe2m0 NECAT_RC_PIPE=2 NECAT_RC3_MIN=80000
run pipe2ww0 NECAT_RC_PIPE=2 NECAT_RC_WW=0
run pipe2min NECAT_RC_PIPE=2 NECAT_RC_PIPE_MIN=100000
