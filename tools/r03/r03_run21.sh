cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" \
            "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH"; do
  i=$((i+1)); rm -rf $O/pmc_$i
  timeout 300 rocprofv3 --pmc $pass -d $O/pmc_$i -o r --output-format csv -- $CMD > $O/pmc_$i.log 2>&1; echo "pmc pass $i rc $?"
done
python tools/make_profiles.py counters $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/r03_sq_counters_b.json; rm -rf $O/pmc_*/
rm -rf $O/prof_stats; rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r --output-format csv -- $CMD > $O/prof_stats.log 2>&1
python tools/make_profiles.py timeline $O/prof_stats $O/r03_round_timeline_b.txt "one bench step kernel by kernel (rocprofv3 --kernel-trace, $CMD; last step with an extension)"
rm -rf $O/prof_stats
