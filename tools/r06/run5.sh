# round 6, call 5: fragments cut inside the checkpoint pass (NECAT_FRAG_FUSE, default on): parity (block level + alternative paths + batches, the full-size md5s of all
# five data shapes, the fuzzers), bench A/B on one box: fused / k_ext_frag in its own launch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py tests/test_gpu_cli_golden.py -q -x -m gpu -k "not drosophila and not human" > $O/run5_parity.txt 2>&1; echo "parity rc $?"; tail -5 $O/run5_parity.txt
for f in 1 0 1 0; do
NECAT_FRAG_FUSE=$f timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-widened --no-pmc > $O/run5_bench_f$f.json 2> $O/run5_bench_f$f.err; echo "bench fuse=$f rc $?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06/run5_bench_f$f.json') if l.startswith('{"metric"')][-1])
print($f, d['ms_per_step'], d['phases_ms_per_step'])
PY
done
