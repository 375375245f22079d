cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
NECAT_FAST16=0 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "test_m4_matches_oracle" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -s -k "test_m4_matches_oracle" > $O/pytest_i.log 2>&1; echo rc $?; grep -v "^  File" $O/pytest_i.log | head -30 | cut -c1-300
