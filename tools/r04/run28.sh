cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
s=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > $O/run28_parity.txt 2>&1; echo "parity rc $? in $(( $(date +%s) - s )) s"; tail -2 $O/run28_parity.txt
s=$(date +%s)
NECAT_RC_PIPE=3 NECAT_RC_PIPE_MIN=64 NECAT_RCWALK=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli_golden.py -q -m gpu -x > $O/run28_parity_pipe3.txt 2>&1; echo "parity (every rc round in 3 pieces) rc $? in $(( $(date +%s) - s )) s"; tail -2 $O/run28_parity_pipe3.txt
for pipe in 1 2 3 4 2 1; do
  NECAT_RC_PIPE=$pipe timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/run28_bench_$pipe.json 2> $O/run28_bench_$pipe.err; echo "bench pipe=$pipe rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r04/run28_bench_$pipe.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'])
PY
done
