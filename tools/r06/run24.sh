# round 6, call 24: a 2 Gbp volume against itself (140 Mb genome x 14.3: one volume of BASELINE configs[3]'s shape) with 1 / 2 whole pairs in flight, arenas warm - what pair lanes
# would give a LONG job of such pairs (run11's 2.5 s project is dominated by its first-touch allocations)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
PIPE_GENOME=140000000 PIPE_COV=14.3 PIPE_Z=20 PIPE_SEED=5 timeout 2400 python tools/r06/pipe2.py 4 1 2 1 2 > $O/run24_pipe_2gbp.txt 2>&1; echo "rc $?"; tail -6 $O/run24_pipe_2gbp.txt
PIPE_JOB=0 PIPE_GENOME=140000000 PIPE_COV=14.3 PIPE_Z=20 PIPE_SEED=5 timeout 2400 python tools/r06/pipe2.py 4 1 2 1 > $O/run24_pipe_2gbp_j0.txt 2>&1; echo "rc $?"; tail -5 $O/run24_pipe_2gbp_j0.txt
