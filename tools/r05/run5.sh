# round 5, call 5: k_rcwalk3 for the big launches only (NECAT_RC3_MIN), wave priorities, against the round-4 default
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run5_$n.json 2> $O/run5_$n.err; echo "bench $n ($*) rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run5_$n.json') if l.startswith('{"metric"')][-1])
print('   ', d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['phases_ms_per_step']['rcwalk_kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
}
run base NECAT_RC_WW=1
run min160 NECAT_RC3_MIN=160000
run min160p9 NECAT_RC3_MIN=160000 NECAT_RC_PRIO=9
run min160p8 NECAT_RC3_MIN=160000 NECAT_RC_PRIO=8
run min120 NECAT_RC3_MIN=120000
run base2 NECAT_RC_WW=1
run p9 NECAT_RC_PRIO=9
run ww3 NECAT_RC_WW=3
