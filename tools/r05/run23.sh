# round 5, call 23: lane 1 on two streams of its own (eight per context), the copy stream made only for the calls that use it: parity of the lane modes, the cut step as
# first / second context, the bench line (its A/B now after the bench's own context is closed), yeast
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cns.py -q -x -k "several_batches or map_pair_equals or capped_band or cns" > $O/run23_parity.txt 2>&1; echo "parity rc $?"; tail -2 $O/run23_parity.txt
timeout 1200 python -m pytest tests/test_gpu_full_size.py -q -x -k "ecoli or yeast" > $O/run23_full.txt 2>&1; echo "full-size rc $?"; tail -2 $O/run23_full.txt
for o in first second; do timeout 300 python tools/r05/ab_cut.py $o 2>&1 | grep -v "^$" | tail -9; done > $O/run23_ab.txt; cat $O/run23_ab.txt
timeout 1500 python bench.py --no-cpu-baseline > $O/run23_bench.json 2> $O/run23_bench.err; echo "bench rc $?"; tail -2 $O/run23_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05/run23_bench.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['roofline']['frac'])
print(d.get('two_lanes_one_batch_cut'))
print(d['extra_configs']['configs2_sensitive']['m4_job1'])
PY
