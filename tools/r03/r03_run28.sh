cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 500 python tests/tools/fuzz_parity.py 30 9100 2 > $O/fuzz_parity.txt 2>&1; echo "fuzz_parity rc $?"; tail -3 $O/fuzz_parity.txt
timeout 300 python tests/tools/fuzz_cns.py 6 9200 > $O/fuzz_cns.txt 2>&1; echo "fuzz_cns rc $?"; tail -2 $O/fuzz_cns.txt
timeout 300 python tests/tools/fuzz_asm.py 9300 6 > $O/fuzz_asm.txt 2>&1; echo "fuzz_asm rc $?"; tail -2 $O/fuzz_asm.txt
timeout 300 python tests/tools/fuzz_rm.py 9400 6 > $O/fuzz_rm.txt 2>&1; echo "fuzz_rm rc $?"; tail -2 $O/fuzz_rm.txt
