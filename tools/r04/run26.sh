cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
s=$(date +%s)
NECAT_TB_WAVES=16 timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "oracle or golden or reference or edlib" > $O/run26_parity.txt 2>&1; echo "parity (16 waves) rc $? in $(( $(date +%s) - s )) s"; tail -2 $O/run26_parity.txt
for tw in 16 4 16 4; do
  NECAT_TB_WAVES=$tw timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/run26_bench_$tw.json 2> $O/run26_bench_$tw.err; echo "bench tb_waves=$tw rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r04/run26_bench_$tw.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'])
PY
done
