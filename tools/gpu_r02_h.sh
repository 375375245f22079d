cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q > $O/pytest_h.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest_h.log | cut -c1-300
for v in "NECAT_FAST16=1" "NECAT_FAST16=0" "NECAT_FAST16=1 NECAT_FAST=2"; do echo "== $v"; env $v NECAT_BATCH_CHUNK=200000 timeout 300 python tools/bench_myers.py 200000 2>&1 | tail -1; done | tee $O/ab_fast16.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/bench_h.json 2> $O/bench_h.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02/bench_h.json'))
for k in ('value','ms_per_step','phases_ms_per_step'): print(k, d.get(k))
r=d['roofline']; print({k:r[k] for k in ('frac','computed_frac','useful_over_computed','band_words_per_block','avg_launch_ms','biggest_launch')})
PY
