cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q > $O/pytest_k.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_k.log | cut -c1-300
bash tools/gpu_r02_j.sh
