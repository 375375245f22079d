# round 6, call 2: the new full-size data shapes (repeat families, long-tailed lengths) against the reference's md5s
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_full_size.py -q -x -k "ecoli_repeats or ecoli_longtail" > $O/run2_full.txt 2>&1; echo "full-size rc $?"; tail -30 $O/run2_full.txt
