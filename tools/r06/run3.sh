# round 6, call 3: cold start of oc2pmov as the first GPU work of a fresh box (stage clocks), the chained correction test + the cns / pcan suites after the
# oc2cns pipeline change, the bench with its widened legs (oc2cns_program: partitions, reference wall)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 600 python tools/r06/cold_start.py > $O/run3_cold.txt 2>&1; echo "cold rc $?"; grep "==" $O/run3_cold.txt
timeout 1500 python -m pytest tests/test_gpu_cns_chain.py tests/test_gpu_cns.py tests/test_oc2pcan.py -q -x -m gpu -s > $O/run3_cns.txt 2>&1; echo "cns rc $?"; tail -8 $O/run3_cns.txt; grep "correction chain" $O/run3_cns.txt
timeout 1500 python bench.py --steps 10 --warmup 3 --no-pmc > $O/run3_bench.json 2> $O/run3_bench.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06/run3_bench.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'])
print(json.dumps(d.get('widened_paths',{}).get('oc2cns_program'), indent=1))
print(json.dumps(d.get('oc2pmov_cold_start'), indent=1))
PY
