# round 2, final profile pass: full bench (all legs), kernel stats, per-launch timeline, SQ counters, HBM traffic (PMC)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02/bench_final.json'))
for k in ('value','ms_per_step','phases_ms_per_step','candidates_job0','oc2pmov_cold_start'): print(k, d.get(k))
r=d['roofline']; print({k:r[k] for k in r if k not in ('note',)})
print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:200])
PY
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf $O/prof_stats; rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r --output-format csv -- $CMD > $O/prof_stats.log 2>&1
python tools/make_profiles.py stats $O/prof_stats $O/r02_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $CMD"
cp $(find $O/prof_stats -name "*kernel_trace.csv" | head -1) $O/kt_final.csv; rm -rf $O/prof_stats
cut -c1-70,200-330 $O/r02_kernel_stats.md | head -16
rm -rf $O/prof_fetch $O/prof_write
rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o r --output-format csv -- $CMD > $O/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -o r --output-format csv -- $CMD > $O/prof_write.log 2>&1
python tools/make_profiles.py pmc $O/prof_fetch $O/prof_write $O/r02_pmc_hbm_traffic.json; rm -rf $O/prof_fetch $O/prof_write
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
            "SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES" \
            "GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_WAVES SQ_INSTS_VALU"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $O/pmc_$tag; rocprofv3 --pmc $pass -d $O/pmc_$tag -o r --output-format csv -- $CMD > $O/pmc_$tag.log 2>&1; echo "pmc $tag rc $?"
done
python tools/make_profiles.py counters $O/pmc_SQ_WAVES $O/pmc_SQ_WAIT_ANY $O/pmc_GRBM_GUI_ACTIVE $O/r02_sq_counters.json; rm -rf $O/pmc_*/
