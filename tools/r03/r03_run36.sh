cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_asm_align.py tests/test_gpu_asmpm.py tests/test_gpu_cns.py tests/test_gpu_rm.py -x -q -m gpu > $O/run36_tests.txt 2>&1; echo "tests rc $?"; tail -2 $O/run36_tests.txt
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-widened > $O/run36_bench_$i.json 2> $O/run36_bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r03/run36_bench_$i.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['phases_ms_per_step'])
PY
done
NECAT_TRACE=2 timeout 600 python tests/tools/bench_asmpm.py 400000 15 0.03 > $O/asm_wg.txt 2>&1; grep "asm_align\|reads\|reference" $O/asm_wg.txt | tail -4 | cut -c1-200
timeout 600 python tools/bench_cns.py > $O/cns_wg.txt 2>&1; tail -1 $O/cns_wg.txt | cut -c1-200
