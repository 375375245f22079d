# round 2, GPU call C: fast path of the list-A DP kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_cns.py -x -q > $O/pytest_c.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest_c.log | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/bench_c.json 2> $O/bench_c.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02/bench_c.json'))
for k in ('value','ms_per_step','phases_ms_per_step'): print(k, d.get(k))
print(d['roofline']['avg_launch_ms'], d['roofline']['biggest_launch'])
PY
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf $O/prof_stats; rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r --output-format csv -- $CMD > $O/prof_stats.log 2>&1
python tools/make_profiles.py stats $O/prof_stats $O/r02_kernel_stats_c.md "rocprofv3 --kernel-trace --stats -- $CMD"; rm -rf $O/prof_stats
cut -c1-70,200-330 $O/r02_kernel_stats_c.md | head -16
