# round 4, final profile pass: kernel stats (concurrent streams and NECAT_SERIAL=1), per-launch timeline, HBM traffic (PMC), SQ counters of the dominant
# kernels, the oc2asmpm program's kernel stats, the 2-rank one-device runs of bench.py, then the full default bench line (which reads
# profiles/r04_pmc_hbm_traffic.json made here)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf $O/prof_stats; rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r --output-format csv -- $CMD > $O/prof_stats.log 2>&1
python tools/make_profiles.py stats $O/prof_stats $O/r04_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $CMD"
python tools/make_profiles.py timeline $O/prof_stats $O/r04_round_timeline.txt "one bench step kernel by kernel (rocprofv3 --kernel-trace, $CMD; last step with an extension)"
rm -rf $O/prof_stats
export NECAT_SERIAL=1
rm -rf $O/prof_serial; rocprofv3 --kernel-trace --stats -d $O/prof_serial -o r --output-format csv -- $CMD > $O/prof_serial.log 2>&1
python tools/make_profiles.py stats $O/prof_serial $O/r04_kernel_stats_serial.md "NECAT_SERIAL=1 rocprofv3 --kernel-trace --stats -- $CMD (the four streams of the extension rounds made ONE: every kernel has the chip to itself)"
rm -rf $O/prof_serial
unset NECAT_SERIAL
rm -rf $O/prof_fetch $O/prof_write
rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o r --output-format csv -- $CMD > $O/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -o r --output-format csv -- $CMD > $O/prof_write.log 2>&1
python tools/make_profiles.py pmc $O/prof_fetch $O/prof_write $O/r04_pmc_hbm_traffic.json; rm -rf $O/prof_fetch $O/prof_write
cp $O/r04_pmc_hbm_traffic.json profiles/r04_pmc_hbm_traffic.json
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" \
            "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH"; do
  i=$((i+1)); rm -rf $O/pmc_$i
  timeout 300 rocprofv3 --pmc $pass -d $O/pmc_$i -o r --output-format csv -- $CMD > $O/pmc_$i.log 2>&1; echo "pmc pass $i rc $?"
done
python tools/make_profiles.py counters $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/r04_sq_counters.json; rm -rf $O/pmc_*/
# the same counters with the round-3 walk kernel (every quad walks its own block), for the comparison of NOTES_r04 1
export NECAT_RC_WW=0
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1)); rm -rf $O/pmcq_$i
  timeout 300 rocprofv3 --pmc $pass -d $O/pmcq_$i -o r --output-format csv -- $CMD > $O/pmcq_$i.log 2>&1; echo "pmc (quad walk) pass $i rc $?"
done
python tools/make_profiles.py counters $O/pmcq_1 $O/r04_sq_counters_quad_walk.json; rm -rf $O/pmcq_*/
unset NECAT_RC_WW
python - > $O/asm_gen.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from necat_amd import synth
rs = synth.simulate_reads(5_000_000, 20.0, seed=71, err=0.03, repeat_frac=0.05)
synth.write_volume_dir("/tmp/asm_vols", rs)
PY
A="-n 100 -z 10 -b 2000 -e 0.5 -j 1 -u 0 -a 400"
rm -rf $O/asm_kt; rocprofv3 --kernel-trace --stats -d $O/asm_kt -o r --output-format csv -- necat_amd/csrc/oc2asmpm $A -t 16 /tmp/asm_vols 0 /tmp/mine2.m4 > $O/asm_kt.log 2>&1
python tools/make_profiles.py stats $O/asm_kt $O/r04_asmpm_kernel_stats.md "rocprofv3 --kernel-trace --stats -- oc2asmpm $A -t 16 wrk 0 out (5 Mb genome x 20, 3 % errors, 5 % repeats: 100 Mbp, 12 479 reads)"; rm -rf $O/asm_kt
NECAT_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-widened > $O/r04_bench_2rank_one_device_ipc.json 2> $O/b2.err; echo "2-rank single-volume rc $?"
NECAT_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --parallelism pairs --volumes 3 > $O/r04_bench_pairs_2rank_one_device.json 2> $O/b3.err; echo "2-rank pairs rc $?"
timeout 1500 python bench.py > $O/r04_bench_final.json 2> $O/r04_bench_final.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/r04_bench_final.json').read().strip().splitlines()[-1])
for k in ('value','gbp_aligned_per_s','ms_per_step','phases_ms_per_step'): print(k, d.get(k))
r=d['roofline']; print({k:r[k] for k in ('frac','achieved','traffic','avg_launch_ms','computed_frac','useful_over_computed','k_myers_ck','k_rcwalk2')}); print(r['hbm'])
print(d.get('roofline_index'))
print(d['widened_paths'].get('oc2asmpm'))
print(d['widened_paths'].get('oc2cns_program'))
print(d.get('extra_configs'))
print(d.get('candidates_job0'), d.get('oc2pmov_cold_start'))
print({k: d['cpu_baseline'].get(k) for k in ('value','cores','cpu_quota_cores','mapping_s','t1')})
PY
