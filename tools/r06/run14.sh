# round 6, call 14: the chip's busy fraction inside the timed region at 1 / 3 steps in flight (kernel trace only, long timed region)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
for d in 3 1; do
  CMD="python bench.py --steps 60 --warmup 1 --in-flight $d --no-cpu-baseline --no-widened --no-pmc"
  rm -rf $O/prof_d$d; timeout 900 rocprofv3 --kernel-trace -d $O/prof_d$d -o r --output-format csv -- $CMD > $O/run14_prof_d$d.log 2>&1; echo "prof d=$d rc $?"
  grep -h '"metric"' $O/run14_prof_d$d.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms/step', d['ms_per_step'], 'one', d.get('one_in_flight',{}).get('ms_per_step'))"
  python tools/r06/busy.py $O/prof_d$d > $O/run14_busy_d$d.txt 2>&1; head -4 $O/run14_busy_d$d.txt
  rm -rf $O/prof_d$d
done
