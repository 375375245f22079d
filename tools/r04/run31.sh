cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
tools/ck_microbench | grep "28672 waves"
s=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_asm_align.py tests/test_gpu_asmpm.py -q -m gpu -x > $O/run31_parity.txt 2>&1; echo "parity rc $? in $(( $(date +%s) - s )) s"; tail -2 $O/run31_parity.txt
for i in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/run31_bench_$i.json 2> $O/run31_bench_$i.err; echo "bench rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r04/run31_bench_$i.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'], d['roofline']['frac'])
PY
done
