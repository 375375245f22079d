cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "edlib_blocks or alternative_kernel or onc_align or ultra or long_chains" > $O/run9_parity.txt 2>&1; echo "parity rc $?"; tail -4 $O/run9_parity.txt
timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_cns.py tests/test_gpu_cli_golden.py -q -m gpu -x -k "not drosophila and not multivol" > $O/run9_full.txt 2>&1; echo "full-size rc $?"; tail -3 $O/run9_full.txt
for m in 1 0 1 0; do
  NECAT_RC_FASTB=$m timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-widened > $O/run9_bench_fastb$m.json 2> $O/run9_bench_fastb$m.err
  python3 -c "
import json
d=json.loads(open('$O/run9_bench_fastb$m.json').read().strip().splitlines()[-1]); print('fastb', $m, d['ms_per_step'], d['phases_ms_per_step'])"
done
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
export NECAT_SERIAL=1
rm -rf $O/prof_serial; rocprofv3 --kernel-trace --stats -d $O/prof_serial -o r --output-format csv -- $CMD > $O/prof_serial.log 2>&1
python tools/make_profiles.py stats $O/prof_serial $O/run9_kernel_stats_serial.md "rocprofv3 --kernel-trace --stats -- $CMD (NECAT_SERIAL=1)"; rm -rf $O/prof_serial
unset NECAT_SERIAL
sed -n 10,24p $O/run9_kernel_stats_serial.md | cut -c1-80,330-
