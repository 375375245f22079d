# round 5, call 12: host-side stage ticks of one step (NECAT_TRACE=2): where the seeding stage's 0.5 ms outside its kernels and the extension's host time go
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
NECAT_TRACE=2 timeout 600 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-widened --no-pmc > $O/run12_bench.json 2> $O/run12_trace.err; echo "rc $?"
grep -n "seeding\|\[necat\] extend\|\[necat\] index" $O/run12_trace.err | tail -60
