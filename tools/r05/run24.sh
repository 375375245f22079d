# round 5, call 24: lane 1's streams at the lowest stream priority (a hardware-queue pool of their own): the cut step as first / second context, yeast, the bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for p in 1 0 2; do for o in first second; do echo "NECAT_LANE1_PRIO=$p $o"; NECAT_LANE1_PRIO=$p timeout 300 python tools/r05/ab_cut.py $o 2>&1 | grep "cut" | tail -3; done; done > $O/run24_ab.txt; cat $O/run24_ab.txt
for p in 1 0 2; do
NECAT_LANE1_PRIO=$p timeout 900 python bench.py --genome 12000000 --coverage 50 --seed 11 --scan-window 10 --steps 3 --warmup 1 --no-cpu-baseline --no-widened --no-pmc > $O/run24_y$p.json 2> $O/run24_y$p.err
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run24_y$p.json') if l.startswith('{"metric"')][-1])
print('yeast prio $p', d['ms_per_step'], d['phases_ms_per_step']['extend'])
PY
done
timeout 1500 python bench.py --no-cpu-baseline > $O/run24_bench.json 2> $O/run24_bench.err; echo "bench rc $?"; tail -2 $O/run24_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05/run24_bench.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['roofline']['frac'])
print({k: d['two_lanes_one_batch_cut'].get(k) for k in ('ms_per_step','extend_ms','error')})
print(d['extra_configs']['configs2_sensitive']['m4_job1'])
PY
