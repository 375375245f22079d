# round 6, call 22: the driver's own command on the final tree (wall clock of the whole run, the line's headline)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
s=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/run22_driver_line.json 2> $O/run22_driver_line.err; echo "rc $? in $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06/run22_driver_line.json') if l.startswith('{"metric"')][-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "Gbp/s", d["gbp_aligned_per_s"], "in flight", d["config"]["steps_in_flight"], "one", d["config"]["one_step_at_a_time"])
print("roofline frac", d["roofline"]["frac"], "one", d["roofline"]["frac_one_in_flight"], "timed region", d["roofline"]["timed_region"]["frac"], "traffic", d["roofline"]["traffic"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "cold", d["oc2pmov_cold_start"])
PY
