# round 5, call 9: index split kernels at 512 threads per 4096-record tile against 256 (NECAT_SPLIT_THREADS): parity, exclusive kernel times, -j 0 / -j 1 steps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "index or candidates_match" > $O/run9_parity.txt 2>&1; echo "index parity rc $?"; tail -2 $O/run9_parity.txt
timeout 900 python -m pytest tests/test_gpu_full_size.py -q -x -k "ecoli" > $O/run9_full.txt 2>&1; echo "full-size ecoli rc $?"; tail -2 $O/run9_full.txt
for t in 512 256; do
  rm -rf $O/prof9; NECAT_SPLIT_THREADS=$t NECAT_SERIAL=1 rocprofv3 --kernel-trace --stats -d $O/prof9 -o r --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened --no-pmc --job 0 > $O/run9_prof.log 2>&1
  python tools/make_profiles.py stats $O/prof9 $O/run9_kernel_stats_job0_t$t.md "NECAT_SPLIT_THREADS=$t NECAT_SERIAL=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --job 0"
  echo "threads $t"; grep -E "k_part_hist|k_split|k_subpart|k_slice" $O/run9_kernel_stats_job0_t$t.md | awk -F'|' '{print $3, $5, substr($2,1,50)}'
  rm -rf $O/prof9
  NECAT_SPLIT_THREADS=$t timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run9_bench_t$t.json 2> $O/run9_bench.err; echo "bench rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run9_bench_t$t.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['value'], d['phases_ms_per_step']['index'], d['candidates_job0']['ms_per_step'], d['roofline_index']['frac'])
PY
done
