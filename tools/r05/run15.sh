# round 5, call 15: what the walk kernels would gain if the blocks that keep their ops (15 %) sat together at the back of a list instead of in nearly every workgroup
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for srt in 0 1; do echo "sorted=$srt"; timeout 300 tools/rcwalk_microbench 0.22 15 $srt 2>&1 | grep -E "lean  |lean again|lean, s_setprio|half|tasks past" ; done > $O/run15_sorted.txt; cat $O/run15_sorted.txt
