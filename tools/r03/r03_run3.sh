# round 3: the cooperative oc2asmpm block aligner (tests + timing against the lane-per-alignment kernel and the reference program)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_zz_asm_align.py tests/test_gpu_asmpm.py -m gpu -q --timeout 600 2>&1 | tail -25 > $O/run3_tests.txt
tail -6 $O/run3_tests.txt
for L in 1 0 0 1; do echo "== NECAT_ASM_LANE=$L"; NECAT_ASM_LANE=$L NECAT_TRACE=3 timeout 600 python tests/tools/bench_asmpm.py 400000 15 0.03 2>&1 | grep -v "^\[necat\] index\|seeding\|batch@" | tail -14; done > $O/r03_bench_asmpm.txt 2>&1
cat $O/r03_bench_asmpm.txt
