cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o r --output-format csv -- $CMD > gpurun_out/prof_stats.log 2>&1
tail -1 gpurun_out/prof_stats.log | cut -c1-600
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o r --output-format csv -- $CMD > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -o r --output-format csv -- $CMD > gpurun_out/prof_write.log 2>&1
python tools/make_profiles.py stats gpurun_out/prof_stats gpurun_out/r01_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $CMD"
python tools/make_profiles.py pmc gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/r01_pmc_hbm_traffic.json
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
head -20 gpurun_out/r01_kernel_stats.md
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened 2>/dev/null | tee gpurun_out/bench_now.json | cut -c1-300
