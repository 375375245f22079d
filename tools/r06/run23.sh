# round 6, call 23: configs[2] size (12 Mb x 50, -z 10: four extension batches on the two lanes of ONE context) with 1 / 2 / 3 whole steps in flight - is there anything left to fill?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
PIPE_GENOME=12000000 PIPE_COV=50 PIPE_Z=10 PIPE_SEED=11 timeout 1500 python tools/r06/pipe2.py 6 1 2 3 1 > $O/run23_pipe_yeast.txt 2>&1; echo "rc $?"; tail -6 $O/run23_pipe_yeast.txt
