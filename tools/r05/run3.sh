# round 5, call 3: why k_rcwalk3 v2 is faster alone (617 against 770 us) and slower in the bench (0.56 against 0.51 ms per launch): the microbench at the bench's
# divergence (22 %: distance ~ 108 per block) and with 15 % of the blocks keeping their ops; exclusive per-launch times of both kernels in the bench (NECAT_SERIAL=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 300 tools/rcwalk_microbench 0.22 15 2>&1 | grep -E "blocks:|==|lean  |lean again|ops kept|prio|half" > $O/run3_micro_22_15.txt; cat $O/run3_micro_22_15.txt
for ww in 2 1; do
  rm -rf $O/prof_s$ww
  NECAT_SERIAL=1 NECAT_RC_WW=$ww rocprofv3 --kernel-trace --stats -d $O/prof_s$ww -o r --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened --no-pmc > $O/run3_prof_s$ww.log 2>&1
  python tools/make_profiles.py stats $O/prof_s$ww $O/run3_kernel_stats_serial_ww$ww.md "NECAT_SERIAL=1 NECAT_RC_WW=$ww rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1"
  grep -E "k_rcwalk|k_myers_ck<" $O/run3_kernel_stats_serial_ww$ww.md
  python tools/make_profiles.py timeline $O/prof_s$ww $O/run3_timeline_ww$ww.txt "serial"; grep -E "k_rcwalk[0-9a-z]*<8" $O/run3_timeline_ww$ww.txt | head -34
  rm -rf $O/prof_s$ww
done
