# round 5, call 32: the one-launch tail kernel's size threshold (NECAT_TAIL_FUSED, 512) and the k_rcwalk3 threshold once more on the final code
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run32_$n.json 2> $O/run32_$n.err; echo "bench $n ($*) rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run32_$n.json') if l.startswith('{"metric"')][-1])
print('   ', d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['phases_ms_per_step'].get('fused_tail_launches'), d['roofline']['frac'])
PY
}
run base NECAT_TAIL_FUSED=512
run t1024 NECAT_TAIL_FUSED=1024
run t2048 NECAT_TAIL_FUSED=2048
run t4096 NECAT_TAIL_FUSED=4096
run t256 NECAT_TAIL_FUSED=256
run m130 NECAT_RC3_MIN=130000
run m190 NECAT_RC3_MIN=190000
run base2 NECAT_TAIL_FUSED=512
