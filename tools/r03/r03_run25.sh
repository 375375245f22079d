cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zz_asm_align.py tests/test_gpu_asmpm.py tests/test_gpu_cns.py tests/test_gpu_parity.py -x -q -m gpu > $O/run25_tests.txt 2>&1; echo "tests rc $?"; tail -2 $O/run25_tests.txt
NECAT_TRACE=1 timeout 600 python tests/tools/bench_asmpm.py 400000 15 0.03 > $O/asm_pack.txt 2>&1; grep "asm round" $O/asm_pack.txt | head -9; grep "reads\|reference" $O/asm_pack.txt | tail -2
NECAT_TRACE=1 timeout 900 python tests/tools/bench_asmpm.py 5000000 20 0.03 > $O/asm5m_pack.txt 2>&1; grep "asm round" $O/asm5m_pack.txt | sed -n 5,12p; grep "reads\|reference" $O/asm5m_pack.txt | tail -2
timeout 600 python tools/bench_cns.py > $O/cns_pack.txt 2>&1; tail -2 $O/cns_pack.txt | cut -c1-300
