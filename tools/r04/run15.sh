cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for m in 0 1 0 1; do
  NECAT_STREAM_PRIO=$m timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-widened > $O/run15_bench_prio$m.json 2> $O/run15_bench_prio$m.err
  python3 -c "
import json
d=json.loads(open('$O/run15_bench_prio$m.json').read().strip().splitlines()[-1]); print('stream priority', $m, d['ms_per_step'], d['config'].get('overlaps_per_step'), d['phases_ms_per_step'])"
done
