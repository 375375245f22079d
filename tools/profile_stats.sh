cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o r --output-format csv -- $CMD > gpurun_out/prof_stats.log 2>&1
python tools/make_profiles.py stats gpurun_out/prof_stats gpurun_out/kernel_stats_now.md "rocprofv3 --kernel-trace --stats -- $CMD"
rm -rf gpurun_out/prof_stats
cut -c1-60,200-400 gpurun_out/kernel_stats_now.md | head -40
