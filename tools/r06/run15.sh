# round 6, call 15: wave slots left free beside the issue-bound checkpoint pass (NECAT_CK_LDS: 28 / 24 / 20 of its waves per CU instead of 32) with 3 steps in flight -
# do the HBM-bound kernels of the other steps (index build, seeding) use them?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
for l in 0 4700 5600 7000 0 4700; do
  NECAT_CK_LDS=$l timeout 900 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run15_bench_l$l.json 2> $O/run15_bench_l$l.err; echo "bench ck_lds $l rc $?"
  python - $O/run15_bench_l$l.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
o = d.get("one_in_flight", {})
print("  in flight", d["config"].get("steps_in_flight"), "ms/step", d["ms_per_step"], "| one:", o.get("ms_per_step"), "| j0", d["candidates_job0"]["ms_per_step"])
PY
done
