# round 6, call 12: fast_shw_ckr's unrolled windows (NECAT_CKR_FAST) - block-level parity (both geometries, sorted / unsorted / rolled), the alternative paths, fuzz,
# full-size md5s, then A/B of the bench line at 1 and 3 steps in flight
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "edlib_blocks or onc_align or CKR_FAST or list_b or several_batches" > $O/run12_parity.txt 2>&1; echo "parity rc $?"; tail -3 $O/run12_parity.txt
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -x > $O/run12_fuzz.txt 2>&1; echo "fuzz rc $?"; tail -2 $O/run12_fuzz.txt
timeout 1500 python -m pytest tests/test_gpu_full_size.py -q -x -k "ecoli or yeast" > $O/run12_full.txt 2>&1; echo "full-size rc $?"; tail -3 $O/run12_full.txt
line() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][-1])
o = d.get("one_in_flight", {})
print(sys.argv[1].split("/")[-1], "in flight", d["config"].get("steps_in_flight"), "ms/step", d["ms_per_step"], "| one:", o.get("ms_per_step"), (o.get("phases_ms_per_step") or d["phases_ms_per_step"]))
PY
}
for f in 1 0 1 0; do
  NECAT_CKR_FAST=$f timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-widened --no-pmc > $O/run12_bench_f$f.json 2> $O/run12_bench_f$f.err; echo "bench ckr_fast $f rc $?"
  line $O/run12_bench_f$f.json
done
