cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
export NECAT_TEST_KEEP_VOLS=/tmp/keepvols
s=$(date +%s)
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/run10_suite.txt 2>&1; echo "GPU suite rc $? in $(( $(date +%s) - s )) s"; tail -6 $O/run10_suite.txt
unset NECAT_TEST_KEEP_VOLS
D=/tmp/keepvols/drosophila
if [ -f $D/vol0 ]; then
  mkdir -p /tmp/d00; ln -sf $D/vol0 /tmp/d00/vol0
  n0=$(head -1 $D/volume_names.txt | cut -f3); printf "1\t%s\n" $n0 > /tmp/d00/reads_info.txt; printf "/tmp/d00/vol0\t0\t%s\n" $n0 > /tmp/d00/volume_names.txt
  OPT="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"
  for big in 0 1; do
    NECAT_INDEX_EMIT_BIG=$big NECAT_TRACE=2 necat_amd/csrc/oc2pmov $OPT -j 0 -u 1 -i 1 -t 8 /tmp/d00 0 /tmp/d00_out_t > $O/run10_d00_big$big.out 2> $O/run10_d00_big$big.err
    echo "emit_big=$big: $(grep 'index: events' $O/run10_d00_big$big.err)"; md5sum /tmp/d00_out_t | cut -c1-32
  done
  for job in 0 1; do
    rm -rf $O/prof_d00_$job
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_d00_$job -o r --output-format csv -- necat_amd/csrc/oc2pmov $OPT -j $job -u $((1-job)) -i $((1-job)) -t 8 /tmp/d00 0 /tmp/d00_out_$job > $O/prof_d00_$job.log 2>&1
    python tools/make_profiles.py stats $O/prof_d00_$job $O/r04_drosophila_v0v0_job${job}_kernel_stats.md "rocprofv3 --kernel-trace --stats -- oc2pmov $OPT -j $job ... (volume 0 of the Drosophila-size set against itself: 2.0 Gbp, 249 592 reads)"
    rm -rf $O/prof_d00_$job
  done
  # the whole project through oc2pm, one worker and two workers on the device: wall per mode
  for mode in "-j 0 -u 1 -i 1" "-j 1 -u 0 -i 0"; do for gp in 0 0,0; do
    rm -f $D/pm*.finished
    s=$(date +%s.%N); NECAT_GPUS=$gp necat_amd/csrc/oc2pm $OPT $mode -t 8 $D /tmp/dros_all > $O/run10_oc2pm.out 2> $O/run10_oc2pm.err; e=$(date +%s.%N)
    python3 -c "print('oc2pm $mode NECAT_GPUS=$gp: %.2f s wall' % ($e - $s))"; ls -la /tmp/dros_all | awk '{print $5}'
  done; done
fi
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export NECAT_SEED_CLEAR_KERNEL=1; else unset NECAT_SEED_CLEAR_KERNEL; fi
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-widened > $O/run10_bench_clear$v.json 2> $O/run10_bench_clear$v.err
  python3 -c "
import json
d=json.loads(open('$O/run10_bench_clear$v.json').read().strip().splitlines()[-1]); print('separate clear kernel', $v, d['ms_per_step'], d['phases_ms_per_step']['seed'], d['candidates_job0']['ms_per_step'])"
done
