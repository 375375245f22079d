# round 5, first GPU call: k_rcwalk3 (32-diagonal records) - bit-equality with k_rcwalk2w on 221 k synthetic blocks + per-launch time (tools/rcwalk_microbench, also
# built for 7 and 6 waves per SIMD), block-level parity of every walk kernel against the oracle, the alternative paths, E. coli full-size md5s, bench A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 300 tools/rcwalk_microbench > $O/run1_micro.txt 2>&1; echo "microbench rc $?"; head -40 $O/run1_micro.txt
for w in 7 6; do timeout 200 tools/rcwalk_microbench_w$w 2>&1 | grep -E "k_rcwalk3 |==" | head -12 > $O/run1_micro_w$w.txt; echo "w$w"; head -8 $O/run1_micro_w$w.txt; done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "edlib_blocks or alternative_kernel" > $O/run1_parity.txt 2>&1; echo "parity rc $?"; tail -3 $O/run1_parity.txt
timeout 900 python -m pytest tests/test_gpu_full_size.py -q -x -k "ecoli" > $O/run1_full.txt 2>&1; echo "full-size ecoli rc $?"; tail -3 $O/run1_full.txt
for ww in 2 1 2 1; do
  NECAT_RC_WW=$ww timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run1_bench_ww$ww.json 2> $O/run1_bench_ww$ww.err; echo "bench ww=$ww rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run1_bench_ww$ww.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['phases_ms_per_step']['rcwalk_kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('roofline_seed',{}).get('achieved'))
PY
done
