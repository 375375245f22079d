cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf $O/prof_t; NECAT_SERIAL=1 rocprofv3 --kernel-trace --stats -d $O/prof_t -o r --output-format csv -- $CMD > $O/prof_t.log 2>&1
python tools/make_profiles.py timeline $O/prof_t $O/run39_round_timeline_serial.txt "one bench step kernel by kernel, NECAT_SERIAL=1"
rm -rf $O/prof_t
