# round 5, call 17: the two lanes started at once (every round has ~ 0.3 ms in which its kernels drain and the next ones ramp up - run16: 110 k-block rounds take 1.0 ms,
# 220 k-block ones 1.65): split and order of the two batches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r05; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run17_$n.json 2> $O/run17_$n.err; echo "bench $n ($*) rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run17_$n.json') if l.startswith('{"metric"')][-1])
print('   ', d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
}
run ov0 NECAT_EXT_OVERLAP=0
run p100s50 NECAT_EXT_OVERLAP_PCT=100
run p100s30 NECAT_EXT_OVERLAP_PCT=100 NECAT_EXT_OVERLAP_SPLIT=30
run p100s20 NECAT_EXT_OVERLAP_PCT=100 NECAT_EXT_OVERLAP_SPLIT=20
run p100s70 NECAT_EXT_OVERLAP_PCT=100 NECAT_EXT_OVERLAP_SPLIT=70
run p100s50even NECAT_EXT_OVERLAP_PCT=100 NECAT_EXT_ORDER=0
run p100s30m100 NECAT_EXT_OVERLAP_PCT=100 NECAT_EXT_OVERLAP_SPLIT=30 NECAT_RC3_MIN=100000
run p100s50m100 NECAT_EXT_OVERLAP_PCT=100 NECAT_RC3_MIN=100000
run p100s50evenm100 NECAT_EXT_OVERLAP_PCT=100 NECAT_EXT_ORDER=0 NECAT_RC3_MIN=100000
run p100s50evenm0 NECAT_EXT_OVERLAP_PCT=100 NECAT_EXT_ORDER=0 NECAT_RC_WW=2
run b76 NECAT_EXT_OVERLAP_PCT=100 NECAT_BATCH=76288
run b57 NECAT_EXT_OVERLAP_PCT=100 NECAT_BATCH=57216
