cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf $O/prof_kt; rocprofv3 --kernel-trace -d $O/prof_kt -o r --output-format csv -- $CMD > $O/prof_kt.log 2>&1
python tools/make_profiles.py timeline $O/prof_kt $O/r03_round_timeline_rc.txt "one bench step kernel by kernel (rocprofv3 --kernel-trace, $CMD; last step), NECAT_RCWALK=16384"
rm -rf $O/prof_kt
