# round 5, call 28: do the walk kernels' raised wave priorities (NECAT_RC_PRIO) cost the kernels that run beside them?  piped pass / walk and the cut batch without them
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run28_$n.json 2> $O/run28_$n.err; echo "bench $n ($*) rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run28_$n.json') if l.startswith('{"metric"')][-1])
print('   ', d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
}
run base NECAT_RC_PIPE=1
run prio0 NECAT_RC_PRIO=0
run pipe2prio0 NECAT_RC_PIPE=2 NECAT_RC_PRIO=0
run pipe2prio2 NECAT_RC_PIPE=2 NECAT_RC_PRIO=2
run cut NECAT_EXT_OVERLAP_MIN=131072 NECAT_RC3_MIN=130000
run cutprio0 NECAT_EXT_OVERLAP_MIN=131072 NECAT_RC3_MIN=130000 NECAT_RC_PRIO=0
run cutprio2 NECAT_EXT_OVERLAP_MIN=131072 NECAT_RC3_MIN=130000 NECAT_RC_PRIO=2
run cutlane0 NECAT_EXT_OVERLAP_MIN=131072 NECAT_RC3_MIN=130000 NECAT_LANE1_PRIO=0
