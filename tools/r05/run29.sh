# round 5, call 29: the stage clocks of a cold oc2pmov run (NECAT_CLI_TRACE=1, NECAT_TRACE=2)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 600 python tools/r05/cold_start.py > $O/run29_cold.txt 2>&1; echo rc $?; cat $O/run29_cold.txt | cut -c1-200 | tail -150
