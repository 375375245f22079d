# round 6, profile pass: kernel stats of the default bench (4 steps in flight), of one step at a time, and with the extension's streams made one (NECAT_SERIAL=1:
# every kernel alone on the chip); per-launch timeline of one step; the chip's busy fraction and kernel concurrency (tools/r06/busy.py) at 3 / 1 in flight; the 2-rank
# one-device runs; the whole GPU suite; smoke; then the full default bench line (which measures its own HBM traffic: gpurun_out/pmc_live.json)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
CMD3="python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-widened --no-pmc"
CMD1="python bench.py --steps 3 --warmup 1 --in-flight 1 --no-cpu-baseline --no-widened --no-pmc"
rm -rf $O/prof_stats; rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r --output-format csv -- $CMD3 > $O/prof_stats3.log 2>&1
python tools/make_profiles.py stats $O/prof_stats $O/r06_kernel_stats.md "rocprofv3 --kernel-trace --stats -- $CMD3 (the default: 4 steps in flight; a kernel's duration includes what it waits for beside the other steps' kernels)"
python tools/r06/busy.py $O/prof_stats > $O/r06_busy_4_in_flight.txt 2>&1
rm -rf $O/prof_stats
rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r --output-format csv -- $CMD1 > $O/prof_stats1.log 2>&1
python tools/make_profiles.py stats $O/prof_stats $O/r06_kernel_stats_one_in_flight.md "rocprofv3 --kernel-trace --stats -- $CMD1"
python tools/make_profiles.py timeline $O/prof_stats $O/r06_round_timeline.txt "one bench step kernel by kernel (rocprofv3 --kernel-trace, $CMD1; last step with an extension)"
python tools/r06/round_table.py $O/r06_round_timeline.txt > $O/r06_round_table.txt 2>&1
python tools/r06/busy.py $O/prof_stats > $O/r06_busy_1_in_flight.txt 2>&1
rm -rf $O/prof_stats
export NECAT_SERIAL=1
rocprofv3 --kernel-trace --stats -d $O/prof_serial -o r --output-format csv -- $CMD1 > $O/prof_serial.log 2>&1
python tools/make_profiles.py stats $O/prof_serial $O/r06_kernel_stats_serial.md "NECAT_SERIAL=1 rocprofv3 --kernel-trace --stats -- $CMD1 (the four streams of the extension rounds made ONE: every kernel has the chip to itself)"
rm -rf $O/prof_serial
unset NECAT_SERIAL
NECAT_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-widened --no-pmc > $O/r06_bench_2rank_one_device_replicated.json 2> $O/b2.err; echo "2-rank single-volume (index plan) rc $?"
NECAT_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --parallelism pairs --volumes 3 --no-pmc > $O/r06_bench_pairs_2rank_one_device.json 2> $O/b3.err; echo "2-rank pairs rc $?"
s=$(date +%s)
timeout 2700 python -m pytest tests/ -q -m gpu -x > $O/r06_final_gpu_tests.txt 2>&1; echo "GPU suite rc $? in $(( $(date +%s) - s )) s"; tail -6 $O/r06_final_gpu_tests.txt | head -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/r06_smoke.txt
s=$(date +%s)
timeout 1500 python bench.py > $O/r06_bench_final.json 2> $O/r06_bench_final.err; echo "bench rc $? in $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/r06_bench_final.json').read().strip().splitlines()[-1])
for k in ('value','gbp_aligned_per_s','ms_per_step','phases_ms_per_step'): print(k, d.get(k))
print('one_in_flight', {k: d.get('one_in_flight', {}).get(k) for k in ('ms_per_step', 'phases_ms_per_step')})
r=d['roofline']; print({k:r.get(k) for k in ('frac','achieved','traffic','avg_launch_ms','computed_frac','useful_over_computed','k_myers_ck','k_rcwalk','timed_region')}); print(r['hbm'])
print('one roofline', d.get('one_in_flight', {}).get('roofline'))
print(d.get('roofline_index')); print(d.get('roofline_seed'))
print(d['widened_paths'].get('oc2asmpm')); print(d['widened_paths'].get('oc2cns_program'))
print(d.get('extra_configs')); print(d.get('candidates_job0'), d.get('oc2pmov_cold_start'), d.get('end_to_end_with_h2d'))
print({k: d['cpu_baseline'].get(k) for k in ('value','cores','cpu_quota_cores','mapping_s','t1')})
PY
