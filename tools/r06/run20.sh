# round 6, call 20: fast_shw_ckr's unrolled windows only where they do not cost the full blocks' pass its registers (k_myers_ckf; k_myers_ck's ISA is round 5's again):
# parity, the bench at 4 and 1 in flight twice
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "edlib_blocks or onc_align or CKR_FAST or several_batches or list_b" > $O/run20_parity.txt 2>&1; echo "parity rc $?"; tail -2 $O/run20_parity.txt
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_fuzz.py -q -x -k "ecoli or yeast or fuzz" > $O/run20_full.txt 2>&1; echo "full-size + fuzz rc $?"; tail -2 $O/run20_full.txt
for r in 1 2 3; do
  timeout 900 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run20_b.json 2> $O/run20_b.err || echo FAILED
  python - <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/r06/run20_b.json') if l.startswith('{"metric"')][-1])
o = d["roofline"]["one_in_flight"]
print("ms/step", d["ms_per_step"], "| one", o["ms_per_step"], "myers_kernel", o["phases_ms_per_step"]["myers_kernel"], "avg launch", o["avg_launch_ms"], "frac one", o["frac"])
PY
done
