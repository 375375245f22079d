cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zz_asm_align.py tests/test_gpu_asmpm.py -x -q -m gpu > $O/run15_tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/run15_tests.txt
for v in 1 0; do NECAT_ASM_RC=$v NECAT_TRACE=1 timeout 600 python tests/tools/bench_asmpm.py 400000 15 0.03 > $O/asm_rc_$v.txt 2>&1; echo "asm rc=$v: $?"; grep -c "asm round" $O/asm_rc_$v.txt; grep "asm round" $O/asm_rc_$v.txt | head -12; grep "reads\|reference\|asm aligner\|align" $O/asm_rc_$v.txt | tail -6; done
for v in 1 0; do NECAT_ASM_RC=$v NECAT_TRACE=1 timeout 900 python tests/tools/bench_asmpm.py 5000000 20 0.03 > $O/asm5m_rc_$v.txt 2>&1; echo "asm 5M rc=$v: $?"; grep "asm round" $O/asm5m_rc_$v.txt | head -12; grep "reads\|reference" $O/asm5m_rc_$v.txt | tail -3; done
