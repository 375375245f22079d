# round 6, call 9: D whole steps side by side on one device (D contexts, D host threads): throughput at depth 1 / 2 / 3, -j 1 and -j 0
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python tools/r06/pipe2.py 20 1 2 3 4 1 2 > $O/run9_pipe_j1.txt 2>&1; echo "j1 rc $?"; cat $O/run9_pipe_j1.txt | tail -8
PIPE_JOB=0 timeout 900 python tools/r06/pipe2.py 20 1 2 3 1 > $O/run9_pipe_j0.txt 2>&1; echo "j0 rc $?"; cat $O/run9_pipe_j0.txt | tail -6
