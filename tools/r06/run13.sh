# round 6, call 13: what the chip does with 1 / 3 steps in flight - kernel trace + HIP API trace of a short bench run, busy.py on both
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
for d in 3 1; do
  CMD="python bench.py --steps 9 --warmup 2 --in-flight $d --no-cpu-baseline --no-widened --no-pmc"
  rm -rf $O/prof_d$d; timeout 900 rocprofv3 --kernel-trace --hip-trace --stats -d $O/prof_d$d -o r --output-format csv -- $CMD > $O/run13_prof_d$d.log 2>&1; echo "prof d=$d rc $?"
  grep -h '"metric"' $O/run13_prof_d$d.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms/step', d['ms_per_step'], 'one', d.get('one_in_flight',{}).get('ms_per_step'))"
  python tools/r06/busy.py $O/prof_d$d > $O/run13_busy_d$d.txt 2>&1; cat $O/run13_busy_d$d.txt
  python tools/make_profiles.py stats $O/prof_d$d $O/run13_kernel_stats_d$d.md "rocprofv3 --kernel-trace --hip-trace --stats -- $CMD"
  du -sh $O/prof_d$d; rm -rf $O/prof_d$d
done
