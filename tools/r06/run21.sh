# round 6, call 21: the checkpoint pass re-reads the previous target window from LDS instead of carrying it in two registers (spills 76 -> 68 bytes, one spill / reload pair
# left inside the unrolled window instead of three): parity, then the pass's launch average against run20's 0.481 ms
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "edlib_blocks or onc_align or several_batches" > $O/run21_parity.txt 2>&1; echo "parity rc $?"; tail -2 $O/run21_parity.txt
timeout 1500 python -m pytest tests/test_gpu_full_size.py -q -x -k "ecoli and not batch" > $O/run21_full.txt 2>&1; echo "full-size rc $?"; tail -2 $O/run21_full.txt
for r in 1 2 3; do
  timeout 900 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run21_b.json 2> $O/run21_b.err || echo FAILED
  python - <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/r06/run21_b.json') if l.startswith('{"metric"')][-1])
o = d["roofline"]["one_in_flight"]
print("ms/step", d["ms_per_step"], "| one", o["ms_per_step"], "myers_kernel", o["phases_ms_per_step"]["myers_kernel"], "avg launch", o["avg_launch_ms"], "frac one", o["frac"])
PY
done
