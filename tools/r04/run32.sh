cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf $O/prof_t; rocprofv3 --kernel-trace --stats -d $O/prof_t -o r --output-format csv -- $CMD > $O/prof_t.log 2>&1
python tools/make_profiles.py stats $O/prof_t $O/run32_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $CMD"
python tools/make_profiles.py timeline $O/prof_t $O/run32_round_timeline.txt "one bench step kernel by kernel"
rm -rf $O/prof_t
