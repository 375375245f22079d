# round 5, final profile pass: kernel stats (concurrent streams and NECAT_SERIAL=1), per-launch timeline, SQ counters of the dominant kernels, the 2-rank
# one-device runs of bench.py (index replicated by necat_index_plan / forced into slices), the whole GPU suite, smoke, then the full default bench line - which
# measures its own HBM traffic (two rocprofv3 --pmc passes inside bench.py: gpurun_out/pmc_live.json -> profiles/r05_pmc_hbm_traffic.json)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened --no-pmc"
rm -rf $O/prof_stats; rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r --output-format csv -- $CMD > $O/prof_stats.log 2>&1
python tools/make_profiles.py stats $O/prof_stats $O/r05_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $CMD"
python tools/make_profiles.py timeline $O/prof_stats $O/r05_round_timeline.txt "one bench step kernel by kernel (rocprofv3 --kernel-trace, $CMD; last step with an extension)"
rm -rf $O/prof_stats
export NECAT_SERIAL=1
rm -rf $O/prof_serial; rocprofv3 --kernel-trace --stats -d $O/prof_serial -o r --output-format csv -- $CMD > $O/prof_serial.log 2>&1
python tools/make_profiles.py stats $O/prof_serial $O/r05_kernel_stats_serial.md "NECAT_SERIAL=1 rocprofv3 --kernel-trace --stats -- $CMD (the four streams of the extension rounds made ONE: every kernel has the chip to itself)"
rm -rf $O/prof_serial
unset NECAT_SERIAL
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" \
            "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH"; do
  i=$((i+1)); rm -rf $O/pmc_$i
  timeout 300 rocprofv3 --pmc $pass -d $O/pmc_$i -o r --output-format csv -- $CMD > $O/pmc_$i.log 2>&1; echo "pmc pass $i rc $?"
done
python tools/make_profiles.py counters $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/r05_sq_counters.json; rm -rf $O/pmc_*/
NECAT_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-widened --no-pmc > $O/r05_bench_2rank_one_device_replicated.json 2> $O/b2.err; echo "2-rank single-volume (index plan) rc $?"
NECAT_BENCH_ONE_DEVICE=1 NECAT_INDEX_SHARD=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-widened --no-pmc > $O/r05_bench_2rank_one_device_slices.json 2> $O/b2s.err; echo "2-rank single-volume (slices) rc $?"
NECAT_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --parallelism pairs --volumes 3 --no-pmc > $O/r05_bench_pairs_2rank_one_device.json 2> $O/b3.err; echo "2-rank pairs rc $?"
s=$(date +%s)
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/r05_final_gpu_tests.txt 2>&1; echo "GPU suite rc $? in $(( $(date +%s) - s )) s"; tail -6 $O/r05_final_gpu_tests.txt | head -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/r05_smoke.txt
timeout 1500 python bench.py > $O/r05_bench_final.json 2> $O/r05_bench_final.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/r05_bench_final.json').read().strip().splitlines()[-1])
for k in ('value','gbp_aligned_per_s','ms_per_step','phases_ms_per_step'): print(k, d.get(k))
r=d['roofline']; print({k:r[k] for k in ('frac','achieved','traffic','avg_launch_ms','computed_frac','useful_over_computed','k_myers_ck','k_rcwalk')}); print(r['hbm'])
print(d.get('roofline_index')); print(d.get('roofline_seed'))
print(d['widened_paths'].get('oc2asmpm')); print(d['widened_paths'].get('oc2cns_program'))
print(d.get('extra_configs')); print(d.get('candidates_job0'), d.get('oc2pmov_cold_start'), d.get('end_to_end_with_h2d'))
print({k: d['cpu_baseline'].get(k) for k in ('value','cores','cpu_quota_cores','mapping_s','t1')})
PY
