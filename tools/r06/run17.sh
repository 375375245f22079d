# round 6, call 17: repetition of the new concurrent paths - contexts on threads, pair lanes, the in-flight bench line (different D), the multi-volume project with 3 lanes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
: > $O/run17_stress.txt
fail=0
for i in $(seq 1 12); do
  timeout 600 python -m pytest tests/test_gpu_cli_golden.py -q -x -k "lanes or threads" >> $O/run17_stress.txt 2>&1 || { fail=$((fail+1)); echo "rep $i FAILED"; }
done
echo "threads / lanes tests: 12 repetitions, $fail failed"
for d in 2 3 4 6 3 3; do
  timeout 600 python bench.py --steps 40 --warmup 1 --in-flight $d --no-cpu-baseline --no-widened --no-pmc > $O/run17_b.json 2> $O/run17_b.err || echo "bench D=$d FAILED"
  python - $d <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/r06/run17_b.json') if l.startswith('{"metric"')][-1])
print("D", sys.argv[1], "ms/step", d["ms_per_step"], "records/step", d["config"]["overlaps_per_step"], "one", d.get("roofline", {}).get("one_in_flight", {}).get("ms_per_step"))
PY
done
NECAT_PAIR_LANES=3 timeout 1500 python -m pytest tests/test_gpu_full_size.py -q -x -k "multivolume_project" > $O/run17_multivol_lanes3.txt 2>&1; echo "multivol project, 3 lanes rc $?"; tail -2 $O/run17_multivol_lanes3.txt
timeout 900 python -m pytest tests/test_gpu_pairs.py -q -x -k "in_flight" > $O/run17_bench_test.txt 2>&1; echo "bench in-flight test rc $?"; tail -2 $O/run17_bench_test.txt
