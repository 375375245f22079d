cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
rm -rf $O/asm_kt; rocprofv3 --kernel-trace --stats -d $O/asm_kt -o r --output-format csv -- python tests/tools/bench_asmpm.py 400000 15 0.03 > $O/asm_kt.log 2>&1
python tools/make_profiles.py stats $O/asm_kt $O/r03_asmpm_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python tests/tools/bench_asmpm.py 400000 15 0.03"; rm -rf $O/asm_kt
