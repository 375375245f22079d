# round 6, call 2: the new full-size data shapes (repeat families, 6 % errors, long-tailed lengths) against the reference's md5s; the multi-rank
# one-device tests after the plan-agreement / first-contact changes; one batch cut in two at several split points (lane 0 = the longest chains)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_full_size.py -q -x -k "ecoli_repeats or ecoli_err6 or ecoli_longtail" > $O/run2_full.txt 2>&1; echo "full-size rc $?"; tail -30 $O/run2_full.txt
timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_pairs.py tests/test_abi.py -q -x -m gpu > $O/run2_shard.txt 2>&1; echo "shard rc $?"; tail -5 $O/run2_shard.txt
for sp in 0 5 10 20 35; do
  if [ $sp = 0 ]; then E=""; else E="NECAT_EXT_OVERLAP_MIN=100000 NECAT_EXT_OVERLAP_SPLIT=$sp"; fi
  env $E timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run2_bench_sp$sp.json 2> $O/run2_bench_sp$sp.err; echo "bench split=$sp rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06/run2_bench_sp$sp.json') if l.startswith('{"metric"')][-1])
print($sp, d['ms_per_step'], d['phases_ms_per_step']['extend'])
PY
done
