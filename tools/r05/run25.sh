# round 5, call 25: does the timed region still see the device's clocks ramp up?  (tools/r05/ab_cut.py: the first passes of a process 43 - 44 ms, later ones 37 - 38)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for w in 3 10 25 50; do for k in 10 30; do
timeout 600 python bench.py --steps $k --warmup $w --no-cpu-baseline --no-widened --no-pmc > $O/run25_w${w}_k$k.json 2> /dev/null
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run25_w${w}_k$k.json') if l.startswith('{"metric"')][-1])
print('warmup $w steps $k:', d['ms_per_step'], d['phases_ms_per_step']['index'], d['phases_ms_per_step']['seed'], d['phases_ms_per_step']['extend'], d['roofline']['frac'])
PY
done; done
rocm-smi --showclocks 2>/dev/null | head -20
