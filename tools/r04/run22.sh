cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
s=$(date +%s)
timeout 1500 python -m pytest tests/ -q -m gpu > $O/run22_suite.txt 2>&1; echo "GPU suite rc $? in $(( $(date +%s) - s )) s"; tail -3 $O/run22_suite.txt
