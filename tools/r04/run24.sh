cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
export NECAT_SERIAL=1
for post in 1 0; do
  rm -rf $O/prof_s$post; NECAT_CK_POST=$post rocprofv3 --kernel-trace --stats -d $O/prof_s$post -o r --output-format csv -- $CMD > $O/prof_s$post.log 2>&1
  python tools/make_profiles.py stats $O/prof_s$post $O/run24_kernel_stats_serial_post$post.md "NECAT_SERIAL=1 NECAT_CK_POST=$post rocprofv3 --kernel-trace --stats -- $CMD"
  rm -rf $O/prof_s$post
  grep "k_myers_ck<8, 16, true>\|k_rcwalk2w<8, 16" $O/run24_kernel_stats_serial_post$post.md | sed 's/(necat::BlockItem[^|]*|/|/' | cut -c1-160
done
unset NECAT_SERIAL
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR"; do
  rm -rf $O/pmc_p; NECAT_SERIAL=1 timeout 300 rocprofv3 --pmc $pass -d $O/pmc_p -o r --output-format csv -- $CMD > $O/pmc_p.log 2>&1; echo "pmc rc $?"
done
python tools/make_profiles.py counters $O/pmc_p $O/run24_sq_counters_post1.json; rm -rf $O/pmc_p
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/run24_sq_counters_post1.json'))
for k,v in d.items():
    if 'k_myers_ck<8, 16, true>' in k: print(json.dumps(v))
PY
