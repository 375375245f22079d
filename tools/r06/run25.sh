# round 6, call 25: the extension's batches on 2 / 3 / 4 lanes of one context (NECAT_EXT_LANES): parity (several-batches cases, yeast md5s), configs[2] size and the E. coli line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "several_batches" > $O/run25_parity.txt 2>&1; echo "parity rc $?"; tail -2 $O/run25_parity.txt
for l in 2 3 4; do
  NECAT_EXT_LANES=$l timeout 1500 python -m pytest tests/test_gpu_full_size.py -q -x -k "yeast or repeats" > $O/run25_full_l$l.txt 2>&1; echo "lanes $l full-size rc $?"; tail -1 $O/run25_full_l$l.txt
done
for l in 2 3 4 2 3 4; do
  NECAT_EXT_LANES=$l PIPE_GENOME=12000000 PIPE_COV=50 PIPE_Z=10 PIPE_SEED=11 timeout 900 python tools/r06/pipe2.py 6 1 > $O/run25_yeast_l$l.txt 2>&1; echo -n "lanes $l: "; tail -1 $O/run25_yeast_l$l.txt
done
for l in 2 3; do
  NECAT_EXT_LANES=$l PIPE_GENOME=140000000 PIPE_COV=14.3 PIPE_Z=20 PIPE_SEED=5 timeout 1500 python tools/r06/pipe2.py 4 1 > $O/run25_2gbp_l$l.txt 2>&1; echo -n "2 Gbp, lanes $l: "; tail -1 $O/run25_2gbp_l$l.txt
done
