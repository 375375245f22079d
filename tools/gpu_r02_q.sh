cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_cli_golden.py tests/test_abi.py -x -q 2>&1 | tail -3
python tools/cli_trace.py 0 2>&1 | tail -24
