cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
NECAT_TRACE=2 timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-widened --no-pmc > $O/run13_bench.json 2> $O/run13_trace.err; echo "rc $?"
grep -n "extend rounds\|filter\|result block\|copy to host" $O/run13_trace.err | tail -40
