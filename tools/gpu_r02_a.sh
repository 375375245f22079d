# round 2, GPU call A: parity after the round-loop rewrite, bench, VALU microbench, kernel stats, SQ counters
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc $?"; cut -c1-700 $O/bench_a.json
NECAT_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-widened > /dev/null 2> $O/trace_rounds.txt; tail -45 $O/trace_rounds.txt | cut -c1-160
./tools/valu_microbench > $O/valu_microbench.txt 2>&1; cat $O/valu_microbench.txt
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf $O/prof_stats; rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r --output-format csv -- $CMD > $O/prof_stats.log 2>&1
python tools/make_profiles.py stats $O/prof_stats $O/r02_kernel_stats_a.md "rocprofv3 --kernel-trace --stats -- $CMD"; rm -rf $O/prof_stats
cut -c1-70,200-330 $O/r02_kernel_stats_a.md | head -24
rocprofv3 -L > $O/counters_list.txt 2>&1
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
            "SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES" \
            "GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_VALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $O/pmc_$tag; rocprofv3 --pmc $pass -d $O/pmc_$tag -o r --output-format csv -- $CMD > $O/pmc_$tag.log 2>&1; echo "pmc $tag rc $?"
done
python tools/make_profiles.py counters $O/pmc_SQ_WAVES $O/pmc_SQ_WAIT_ANY $O/pmc_GRBM_GUI_ACTIVE $O/r02_sq_counters_a.json; rm -rf $O/pmc_*/
head -c 3000 $O/r02_sq_counters_a.json
