cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli_golden.py tests/test_gpu_cns.py -q -m gpu -x > $O/run8_parity.txt 2>&1; echo "parity rc $?"; tail -4 $O/run8_parity.txt
timeout 600 python -m pytest tests/test_gpu_full_size.py -q -m gpu -x -k "ecoli or yeast or identical" > $O/run8_full.txt 2>&1; echo "full-size rc $?"; tail -3 $O/run8_full.txt
for m in 1 0 1; do
  NECAT_RC_MERGE=$m timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-widened > $O/run8_bench_merge$m.json 2> $O/run8_bench_merge$m.err
  python3 -c "
import json
d=json.loads(open('$O/run8_bench_merge$m.json').read().strip().splitlines()[-1]); print('merge', $m, d['ms_per_step'], d['phases_ms_per_step'])"
done
free -g | head -2; nproc
export NECAT_TEST_KEEP_VOLS=/tmp/keepvols
s=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_full_size.py -q -m gpu -x -k "drosophila" > $O/run8_dros.txt 2>&1; echo "drosophila rc $? in $(( $(date +%s) - s )) s"; tail -15 $O/run8_dros.txt
D=/tmp/keepvols/drosophila
if [ -f $D/vol0 ]; then
  mkdir -p /tmp/d00; ln -sf $D/vol0 /tmp/d00/vol0
  n0=$(head -1 $D/volume_names.txt | cut -f3); printf "1\t%s\n" $n0 > /tmp/d00/reads_info.txt; printf "/tmp/d00/vol0\t0\t%s\n" $n0 > /tmp/d00/volume_names.txt
  OPT="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"
  for job in 0 1; do
    rm -rf $O/prof_d00_$job
    s=$(date +%s.%N)
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_d00_$job -o r --output-format csv -- necat_amd/csrc/oc2pmov $OPT -j $job -u $((1-job)) -i $((1-job)) -t 8 /tmp/d00 0 /tmp/d00_out_$job > $O/prof_d00_$job.log 2>&1
    e=$(date +%s.%N); python3 -c "print('oc2pmov (0,0) -j $job under rocprofv3: %.1f s' % ($e - $s))"; ls -la /tmp/d00_out_$job
    python tools/make_profiles.py stats $O/prof_d00_$job $O/r04_drosophila_v0v0_job${job}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- oc2pmov $OPT -j $job ... (volume 0 of the Drosophila-size set against itself: 2.0 Gbp, 249 592 reads)"
    rm -rf $O/prof_d00_$job
  done
  NECAT_TRACE=2 necat_amd/csrc/oc2pmov $OPT -j 0 -u 1 -i 1 -t 8 /tmp/d00 0 /tmp/d00_out_t > $O/run8_d00_trace.out 2> $O/run8_d00_trace.err; tail -30 $O/run8_d00_trace.err
fi
