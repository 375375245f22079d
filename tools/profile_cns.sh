cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CMD="python tools/bench_cns.py"
rm -rf gpurun_out/prof_cns
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cns -o r --output-format csv -- $CMD > gpurun_out/prof_cns.log 2>&1
python tools/make_profiles.py stats gpurun_out/prof_cns gpurun_out/r01_cns_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $CMD   (index build + seeding once, then 3 x necat_cns_extension_batch)"
rm -rf gpurun_out/prof_cns
cut -c1-70,200-330 gpurun_out/r01_cns_kernel_stats.md | head -30
tail -1 gpurun_out/prof_cns.log | cut -c1-300
