cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
s=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py -q -m gpu -x -k "index or shard or oracle" > $O/run37_parity.txt 2>&1; echo "index/shard parity rc $? in $(( $(date +%s) - s )) s"; tail -2 $O/run37_parity.txt
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf $O/prof_t; NECAT_SERIAL=1 rocprofv3 --kernel-trace --stats -d $O/prof_t -o r --output-format csv -- $CMD > $O/prof_t.log 2>&1
python tools/make_profiles.py stats $O/prof_t $O/run37_kernel_stats_serial.md "NECAT_SERIAL=1 rocprofv3 --kernel-trace --stats -- $CMD"
rm -rf $O/prof_t
grep "k_part_hist\|k_split_bases\|k_split_recs\|k_subpart\|k_slice_count\|k_slice_emit" $O/run37_kernel_stats_serial.md | sed 's/(necat::[^|]*|/|/; s/(unsigned[^|]*|/|/' | cut -c1-120
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/run37_bench.json 2> $O/run37_bench.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r04/run37_bench.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'], d.get('candidates_job0',{}).get('ms_per_step'))
PY
