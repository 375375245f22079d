cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for dbg in 0 2 4 0; do
  NECAT_SERIAL=1 NECAT_RC_DBG=$dbg timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-widened > $O/run12_bench_dbg$dbg.json 2> $O/run12_bench_dbg$dbg.err
  python3 -c "
import json
d=json.loads(open('$O/run12_bench_dbg$dbg.json').read().strip().splitlines()[-1]); print('serial, rc_dbg', $dbg, d['ms_per_step'], d['phases_ms_per_step']['rcwalk_kernel'], d['phases_ms_per_step']['extend'])"
done
