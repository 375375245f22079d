# round 5, call 35: block-level and program-level parity with k_rcwalk3 chosen explicitly (NECAT_RC_WW=2, both record widths; NECAT_RC3_MIN=64) - the default takes it
# from 160 k blocks up only, which the 300-block cases never reach; the hook's cross-check context is now made per knob environment
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "edlib_blocks or alternative_kernel or coop_equals or product_library" > $O/run35_parity.txt 2>&1; echo "parity rc $?"; tail -5 $O/run35_parity.txt | cut -c1-300
