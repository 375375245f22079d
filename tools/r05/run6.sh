# round 5, call 6: BASELINE configs[4] as a stated subset (3 Gb genome at the 30x rate, volumes 0 - 2 of 45: 3 x 2 Gbp) - the oc2pm program against the reference
# binary's md5s (tests/test_gpu_full_size.py[human_subset]), then the same volume files through bench.py's configs4 step (index of a near-all-distinct 2 Gbp
# volume, per-pair -j 0 / -j 1 times); parity of the walk kernels with the hybrid default
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
export NECAT_TEST_KEEP_VOLS=/tmp/keepvols; mkdir -p $NECAT_TEST_KEEP_VOLS
s=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_full_size.py -q -x -k "human_subset" > $O/run6_human.txt 2>&1; echo "human_subset rc $? in $(( $(date +%s) - s )) s"; tail -5 $O/run6_human.txt
ls -la /tmp/keepvols/human_subset | head
s=$(date +%s)
NECAT_CONFIG4_VOLS=/tmp/keepvols/human_subset timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened --no-pmc --config4-genome 3000000000 > $O/run6_bench_config4.json 2> $O/run6_bench_config4.err; echo "bench config4 rc $? in $(( $(date +%s) - s )) s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05/run6_bench_config4.json') if l.startswith('{"metric"')][-1])
print(json.dumps(d.get('extra_configs',{}).get('configs4_human_subset'), indent=1)[:6000])
print(d['ms_per_step'], d.get('roofline_seed'))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "edlib_blocks or alternative_kernel" > $O/run6_parity.txt 2>&1; echo "parity rc $?"; tail -3 $O/run6_parity.txt
