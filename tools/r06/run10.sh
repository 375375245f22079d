# round 6, call 10: steps in flight (bench.py --in-flight) and pair lanes (NECAT_PAIR_LANES) - the new tests, the bench line at D = 1 .. 4, hardware queues, and
# BASELINE configs[3] at its real size (5.6 Gbp in 3 volumes, 6 pairs) through one oc2pm worker at 1 / 2 / 3 lanes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_cli_golden.py -q -x > $O/run10_cli_tests.txt 2>&1; echo "cli tests rc $?"; tail -3 $O/run10_cli_tests.txt
line() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
o = d.get("one_in_flight", {})
print(sys.argv[1].split("/")[-1], "in flight", d["config"].get("steps_in_flight"), "ms/step", d["ms_per_step"], "value", d["value"], "| one:", o.get("ms_per_step"), "| frac", d["roofline"]["frac"],
      d["roofline"].get("timed_region", {}).get("frac"), (o.get("roofline") or {}).get("frac"), "| j0", d.get("candidates_job0", {}).get("ms_per_step"), d.get("candidates_job0", {}).get("one_in_flight_ms_per_step"))
PY
}
for d in 3 1 2 4 3; do
  timeout 900 python bench.py --steps 20 --warmup 5 --in-flight $d --no-cpu-baseline --no-widened --no-pmc > $O/run10_bench_d$d.json 2> $O/run10_bench_d$d.err; echo "bench in-flight $d rc $?"
  line $O/run10_bench_d$d.json
done
for q in 4 16 24; do
  GPU_MAX_HW_QUEUES=$q timeout 900 python bench.py --steps 20 --warmup 5 --in-flight 3 --no-cpu-baseline --no-widened --no-pmc > $O/run10_bench_q$q.json 2> $O/run10_bench_q$q.err; echo "bench queues $q rc $?"
  line $O/run10_bench_q$q.json
done
python - > $O/run10_gen.txt 2>&1 <<'PY'
import os, sys, json, time
sys.path.insert(0, os.getcwd())
from necat_amd import synth
g = json.load(open("tests/golden/drosophila_full_reference.json"))["generator"]
rs = synth.simulate_reads(g["genome"], g["coverage"], seed=g["seed"], err=g["err"])
synth.write_volume_dir_cuts("/tmp/dros", rs, g["cuts"])
PY
D=/tmp/dros
OPT="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"
for rep in 1 2; do
for ln in 1 2 3; do
  rm -f $D/pm*.finished
  s=$(date +%s.%N); NECAT_PAIR_LANES=$ln NECAT_GPUS=0 necat_amd/csrc/oc2pm $OPT -j 1 -u 0 -i 0 -t 16 $D /tmp/dros_all > $O/run10_oc2pm.out 2> $O/run10_oc2pm_${rep}_$ln.err; e=$(date +%s.%N)
  python3 -c "print('rep $rep oc2pm -j 1, NECAT_PAIR_LANES=$ln: %.2f s wall' % ($e - $s))"
  sort /tmp/dros_all | md5sum | cut -c1-12
done; done
for ln in 1 2 3; do
  rm -f $D/pm*.finished
  s=$(date +%s.%N); NECAT_PAIR_LANES=$ln NECAT_GPUS=0 necat_amd/csrc/oc2pm $OPT -j 0 -u 1 -i 1 -t 16 $D /tmp/dros_can > $O/run10_oc2pm0.out 2> $O/run10_oc2pm0_$ln.err; e=$(date +%s.%N)
  python3 -c "print('oc2pm -j 0, NECAT_PAIR_LANES=$ln: %.2f s wall' % ($e - $s))"
  md5sum /tmp/dros_can | cut -c1-12
done
python3 -c "
import json; g=json.load(open('tests/golden/drosophila_full_reference.json')); print('golden sorted md5', g['m4_text_sorted_md5'][:12], g['m4_records'])"
wc -l /tmp/dros_all
