# round 5, call 19: one E. coli-size batch cut in two, both lanes at once: the share of the first (longest chains) batch, the k_rcwalk3 size threshold beside it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=8 NECAT_EXT_OVERLAP_PCT=100
O=gpurun_out/r05; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run19_$n.json 2> $O/run19_$n.err; echo "bench $n ($*) rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run19_$n.json') if l.startswith('{"metric"')][-1])
print('   ', d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
}
run ov0 NECAT_EXT_OVERLAP=0
run s10 NECAT_EXT_OVERLAP_SPLIT=10
run s15 NECAT_EXT_OVERLAP_SPLIT=15
run s20 NECAT_EXT_OVERLAP_SPLIT=20
run s25 NECAT_EXT_OVERLAP_SPLIT=25
run s15m130 NECAT_EXT_OVERLAP_SPLIT=15 NECAT_RC3_MIN=130000
run s20m130 NECAT_EXT_OVERLAP_SPLIT=20 NECAT_RC3_MIN=130000
run s25m130 NECAT_EXT_OVERLAP_SPLIT=25 NECAT_RC3_MIN=130000
run s20m100 NECAT_EXT_OVERLAP_SPLIT=20 NECAT_RC3_MIN=100000
run s20p97 NECAT_EXT_OVERLAP_SPLIT=20 NECAT_EXT_OVERLAP_PCT=97
run s20again NECAT_EXT_OVERLAP_SPLIT=20
run ov0again NECAT_EXT_OVERLAP=0
