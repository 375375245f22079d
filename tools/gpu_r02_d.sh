# round 2, GPU call D: new walk (k_traceback) + A/B of the DP fast path on synthetic full blocks
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_cns.py -x -q > $O/pytest_d.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest_d.log | cut -c1-300
for v in "NECAT_FAST=1 NECAT_WALK=1" "NECAT_FAST=0 NECAT_WALK=0" "NECAT_FAST=2 NECAT_WALK=1"; do echo "== $v"; env $v NECAT_BATCH_CHUNK=200000 timeout 300 python tools/bench_myers.py 200000 2>&1 | tail -2; done | tee $O/ab_myers.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_d.json 2> $O/bench_d.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02/bench_d.json'))
for k in ('value','ms_per_step','phases_ms_per_step','widened_paths'): print(k, d.get(k))
print(d['roofline']['avg_launch_ms'], d['roofline']['biggest_launch'])
PY
NECAT_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-widened > /dev/null 2> $O/trace_rounds_d.txt; grep "round" $O/trace_rounds_d.txt | tail -34 | cut -c1-140
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf $O/prof_stats; rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r --output-format csv -- $CMD > $O/prof_stats.log 2>&1
python tools/make_profiles.py stats $O/prof_stats $O/r02_kernel_stats_d.md "rocprofv3 --kernel-trace --stats -- $CMD"; rm -rf $O/prof_stats
cut -c1-70,200-330 $O/r02_kernel_stats_d.md | head -16
