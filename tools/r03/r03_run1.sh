# round 3, first GPU pass: the new multi-rank / pair-schedule paths, the fused tail kernel, A/B of its threshold
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_parity.py tests/test_gpu_shard.py tests/test_gpu_cli_golden.py tests/test_gpu_cns.py tests/test_oc2pcan.py -m gpu -q --timeout 900 2>&1 | tail -40 > $O/run1_tests.txt
tail -15 $O/run1_tests.txt
for T in 0 640 1024 384; do
  NECAT_TAIL_FUSED=$T timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/ab_tail_$T.json 2> $O/ab_tail_$T.err
  python - <<PY
import json
d=json.loads(open('$O/ab_tail_$T.json').read().strip().splitlines()[-1])
print('TAIL_FUSED=$T', d['ms_per_step'], d['config']['overlaps_per_step'], d['phases_ms_per_step'])
PY
done
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q --timeout 900 2>&1 | tail -15 > $O/run1_full.txt
tail -8 $O/run1_full.txt
