cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
for v in "NECAT_FAST16=0" "NECAT_FAST16=1"; do echo "== $v"; env $v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['phases_ms_per_step'], d['roofline']['biggest_launch'])"; done
NECAT_FAST16=1 NECAT_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-widened > /dev/null 2> $O/trace_rounds_j.txt; grep "round" $O/trace_rounds_j.txt | tail -34 | head -14 | cut -c1-140
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf $O/prof_stats; rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r --output-format csv -- $CMD > $O/prof_stats.log 2>&1
python tools/make_profiles.py stats $O/prof_stats $O/r02_kernel_stats_j.md "rocprofv3 --kernel-trace --stats -- $CMD"; rm -rf $O/prof_stats
cut -c1-70,200-330 $O/r02_kernel_stats_j.md | head -14
