# round 5, call 4: k_rcwalk3p (recompute and walk side by side, predicted entries): equality with k_rcwalk2w + time, alone
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 300 tools/rcwalk_microbench > $O/run4_micro.txt 2>&1; echo "microbench rc $?"; grep -v "^---" $O/run4_micro.txt | head -44
timeout 300 tools/rcwalk_microbench 0.22 15 2>&1 | grep -E "blocks:|==|lean  |lean again|ops kept|prio|half|ONE|one workgroup" > $O/run4_micro_22_15.txt; cat $O/run4_micro_22_15.txt
