# per-launch kernel trace of one bench step (timeline analysis of the extension rounds)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
rm -rf $O/kt; rocprofv3 --kernel-trace -d $O/kt -o r --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-widened > $O/kt.log 2>&1
ls -la $O/kt; wc -l $O/kt/*kernel_trace.csv
