# round 6, call 4: k_ext_frag with 4 threads per item (parity: block level, alternative paths, E. coli / yeast md5s, the asm aligner's geometries, the CLI goldens),
# bench A/B against call 1 / 3 (39.02 ms), oc2pmov's fast exit + cold start measured first
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 600 python tools/r06/cold_start.py > $O/run4_cold.txt 2>&1; echo "cold rc $?"; grep "==" $O/run4_cold.txt
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_zz_asm_align.py tests/test_gpu_asmpm.py tests/test_gpu_cli_golden.py -q -x -m gpu -k "not drosophila and not human" > $O/run4_parity.txt 2>&1; echo "parity rc $?"; tail -5 $O/run4_parity.txt
for i in 1 2; do
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-widened --no-pmc > $O/run4_bench$i.json 2> $O/run4_bench$i.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06/run4_bench$i.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'])
PY
done
