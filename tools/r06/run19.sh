# round 6, call 19: steps 7 .. 31 of the full blocks' pass unrolled (NECAT_CK_W0): parity, then A/B at 1 and 4 steps in flight
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "edlib_blocks or onc_align or CK_W0 or CKR_FAST or several_batches" > $O/run19_parity.txt 2>&1; echo "parity rc $?"; tail -2 $O/run19_parity.txt
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -q -x -k "ecoli or yeast or fuzz" > $O/run19_full.txt 2>&1; echo "full-size + fuzz rc $?"; tail -2 $O/run19_full.txt
for w in 1 0 1 0; do
  NECAT_CK_W0=$w timeout 900 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run19_b.json 2> $O/run19_b.err || echo FAILED
  python - $w <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/r06/run19_b.json') if l.startswith('{"metric"')][-1])
o = d["roofline"]["one_in_flight"]
print("CK_W0", sys.argv[1], "ms/step", d["ms_per_step"], "| one", o["ms_per_step"], "myers_kernel", o["phases_ms_per_step"]["myers_kernel"], "avg launch", o["avg_launch_ms"])
PY
done
