# round 6, call 16: thresholds tuned for one step at a time, re-swept with 3 steps in flight (the chip is full more of the time: instruction counts weigh more, chains less)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
run() { env "$@" timeout 900 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run16_tmp.json 2> $O/run16_tmp.err; python - "$*" <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/r06/run16_tmp.json') if l.startswith('{"metric"')][-1])
o = d.get("one_in_flight", {})
print("%-50s in flight %d ms/step %.2f | one: %s" % (sys.argv[1], d["config"].get("steps_in_flight"), d["ms_per_step"], o.get("ms_per_step")))
PY
}
run NECAT_X=0
run NECAT_RC3_MIN=64
run NECAT_RC3_MIN=60000
run NECAT_RC3_MIN=100000000
run NECAT_RC_PRIO=0
run NECAT_RC_PRIO=3
run NECAT_TAIL_FUSED=2048
run NECAT_TAIL_FUSED=128
run NECAT_X=0
