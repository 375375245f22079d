cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_asmpm.py -x -q -m gpu > $O/run33_tests.txt 2>&1; echo "tests rc $?"; tail -2 $O/run33_tests.txt
NECAT_TRACE=2 NECAT_CLI_TRACE=1 timeout 900 python tests/tools/bench_asmpm.py 5000000 20 0.03 > $O/asm5m_trace.txt 2>&1; grep -v "asm round" $O/asm5m_trace.txt | tail -40 | cut -c1-220
