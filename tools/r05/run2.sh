# round 5, call 2: k_rcwalk3 v2 (two lanes per block, both words of the pair per lane, lean walker): equality with k_rcwalk2w + time, parity, bench A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 300 tools/rcwalk_microbench > $O/run2_micro.txt 2>&1; echo "microbench rc $?"; head -24 $O/run2_micro.txt
timeout 200 tools/rcwalk_microbench_w4 2>&1 | grep -E "k_rcwalk3 |==" | head -8 > $O/run2_micro_w4.txt; echo "w4"; cat $O/run2_micro_w4.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "edlib_blocks or alternative_kernel" > $O/run2_parity.txt 2>&1; echo "parity rc $?"; tail -3 $O/run2_parity.txt
timeout 900 python -m pytest tests/test_gpu_full_size.py -q -x -k "ecoli" > $O/run2_full.txt 2>&1; echo "full-size ecoli rc $?"; tail -3 $O/run2_full.txt
for ww in 2 1 2 1; do
  NECAT_RC_WW=$ww timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run2_bench_ww$ww.json 2> $O/run2_bench_ww$ww.err; echo "bench ww=$ww rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run2_bench_ww$ww.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['phases_ms_per_step']['rcwalk_kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
done
