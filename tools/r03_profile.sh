# round 3, final profile pass: kernel stats, per-launch timeline, HBM traffic (PMC), SQ counters of the two dominant kernels, the asm aligner's kernel
# stats, the 2-rank one-device runs of bench.py, then the full bench line (which reads profiles/r03_pmc_hbm_traffic.json made here)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf $O/prof_stats; rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r --output-format csv -- $CMD > $O/prof_stats.log 2>&1
python tools/make_profiles.py stats $O/prof_stats $O/r03_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $CMD"
python tools/make_profiles.py timeline $O/prof_stats $O/r03_round_timeline.txt "one bench step kernel by kernel (rocprofv3 --kernel-trace, $CMD; last step with an extension)"
rm -rf $O/prof_stats
rm -rf $O/prof_fetch $O/prof_write
rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o r --output-format csv -- $CMD > $O/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -o r --output-format csv -- $CMD > $O/prof_write.log 2>&1
python tools/make_profiles.py pmc $O/prof_fetch $O/prof_write $O/r03_pmc_hbm_traffic.json; rm -rf $O/prof_fetch $O/prof_write
cp $O/r03_pmc_hbm_traffic.json profiles/r03_pmc_hbm_traffic.json
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" \
            "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH"; do
  i=$((i+1)); rm -rf $O/pmc_$i
  timeout 300 rocprofv3 --pmc $pass -d $O/pmc_$i -o r --output-format csv -- $CMD > $O/pmc_$i.log 2>&1; echo "pmc pass $i rc $?"
done
python tools/make_profiles.py counters $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/r03_sq_counters.json; rm -rf $O/pmc_*/
rm -rf $O/asm_kt; rocprofv3 --kernel-trace --stats -d $O/asm_kt -o r --output-format csv -- python tests/tools/bench_asmpm.py 400000 15 0.03 > $O/asm_kt.log 2>&1
python tools/make_profiles.py stats $O/asm_kt $O/r03_asmpm_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python tests/tools/bench_asmpm.py 400000 15 0.03"; rm -rf $O/asm_kt
NECAT_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-widened > $O/r03_bench_2rank_one_device_ipc.json 2> $O/b2.err; echo "2-rank single-volume rc $?"
NECAT_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --parallelism pairs --volumes 3 > $O/r03_bench_pairs_2rank_one_device.json 2> $O/b3.err; echo "2-rank pairs rc $?"
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --parallelism pairs --volumes 3 > $O/r03_bench_pairs_1rank.json 2> $O/b4.err; echo "1-rank pairs rc $?"
timeout 1500 python bench.py --asmpm-genome 5000000 > $O/r03_bench_final.json 2> $O/r03_bench_final.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03/r03_bench_final.json').read().strip().splitlines()[-1])
for k in ('value','gbp_aligned_per_s','ms_per_step','phases_ms_per_step'): print(k, d.get(k))
r=d['roofline']; print({k:r[k] for k in ('frac','achieved','traffic','avg_launch_ms','computed_frac','useful_over_computed','k_myers_ck','k_rcwalk2')})
print(d.get('roofline_index'))
print(d['widened_paths'].get('oc2asmpm'))
print(d['widened_paths'].get('oc2cns_program'))
print({k: d['cpu_baseline'].get(k) for k in ('value','cores','cpu_quota_cores','mapping_s','t1')})
PY
