cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
export NECAT_SERIAL=1
rm -rf $O/prof_s; rocprofv3 --kernel-trace --stats -d $O/prof_s -o r --output-format csv -- $CMD > $O/prof_s.log 2>&1
python tools/make_profiles.py stats $O/prof_s $O/run27_kernel_stats_serial.md "NECAT_SERIAL=1 rocprofv3 --kernel-trace --stats -- $CMD"
rm -rf $O/prof_s
sed -n 10,40p $O/run27_kernel_stats_serial.md | sed 's/(necat::[^|]*|/|/; s/(necat_candidate[^|]*|/|/; s/(unsigned[^|]*|/|/' | cut -c1-150
