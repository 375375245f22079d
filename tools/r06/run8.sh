# round 6, call 8: NECAT_DEFER - the walk of a block that keeps no ops stops at its run of matches, its successor is planned from that, a resumed walk on stream d counts
# the block's columns / matches beside the next round.  Parity first (smoke, E. coli md5s, block level), then A/B on one box: off / on / on without the resumed walk (timing only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/run8_smoke.txt 2>&1; echo "smoke rc $?"; tail -2 $O/run8_smoke.txt
timeout 1200 python -m pytest tests/test_gpu_full_size.py -q -x -k "ecoli and not batch_cut" > $O/run8_full.txt 2>&1; echo "full-size rc $?"; tail -4 $O/run8_full.txt
for d in 0 1 2 1 0; do
NECAT_DEFER=$d timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-widened --no-pmc > $O/run8_bench_d$d.json 2> $O/run8_bench_d$d.err; echo "bench defer=$d rc $?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r06/run8_bench_d$d.json') if l.startswith('{"metric"')][-1])
print($d, d['ms_per_step'], d['config']['overlaps_per_step'], d['phases_ms_per_step'])
PY
done
