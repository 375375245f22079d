# round 6, call 7: the bench line after the Gbp reduction left the timed loop (A/B is the same binary: 38.55 - 38.78 before), configs[2] size kernel by kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
for i in 1 2; do
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-widened --no-pmc > $O/run7_bench$i.json 2> $O/run7_bench$i.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06/run7_bench$i.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['value'], d['gbp_aligned_per_s'], d['phases_ms_per_step'])
PY
done
timeout 600 python tools/r06/yeast_prof.py 1 2>&1 | tail -4
NECAT_TRACE=2 timeout 600 python tools/r06/yeast_prof.py 0 2>&1 | tail -40 > $O/run7_yeast_job0_trace.txt; tail -14 $O/run7_yeast_job0_trace.txt
rm -rf $O/prof_yeast; timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_yeast -o r --output-format csv -- python tools/r06/yeast_prof.py 1 > $O/prof_yeast.log 2>&1
python tools/make_profiles.py stats $O/prof_yeast $O/r06_yeast_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python tools/r06/yeast_prof.py 1 (configs[2] size, 4 passes)"
rm -rf $O/prof_yeast
head -30 $O/r06_yeast_kernel_stats.md | cut -c1-150
