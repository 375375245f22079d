# round 5, call 8: index build with one 32-base window per 16 hashed positions (k_part_hist, k_split_bases): parity of the index at every k, kernel stats, -j 0 step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "index or candidates_match" > $O/run8_parity.txt 2>&1; echo "index parity rc $?"; tail -3 $O/run8_parity.txt
timeout 900 python -m pytest tests/test_gpu_full_size.py -q -x -k "ecoli" > $O/run8_full.txt 2>&1; echo "full-size ecoli rc $?"; tail -2 $O/run8_full.txt
rm -rf $O/prof8; NECAT_SERIAL=1 rocprofv3 --kernel-trace --stats -d $O/prof8 -o r --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened --no-pmc --job 0 > $O/run8_prof.log 2>&1
python tools/make_profiles.py stats $O/prof8 $O/run8_kernel_stats_job0.md "NECAT_SERIAL=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --job 0"
grep -E "k_part_hist|k_split|k_subpart|k_slice|k_seed|k_bucket|k_pack" $O/run8_kernel_stats_job0.md | awk -F'|' '{print $3, $5, substr($2,1,50)}'
rm -rf $O/prof8
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run8_bench.json 2> $O/run8_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05/run8_bench.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['value'], d['phases_ms_per_step'], d['candidates_job0']['ms_per_step'], d['roofline_index']['frac'], d.get('end_to_end_with_h2d'))
PY
