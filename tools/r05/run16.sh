# round 5, call 16: two lanes for the extension rounds (NECAT_EXT_OVERLAP: batch i + 1's first rounds beside batch i's last, latency-bound ones; one batch of
# >= 131 072 candidates cut in two): parity (several batches in every lane mode, full-size md5s of E. coli = one batch cut in two and yeast = four batches), then A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "several_batches or map_pair_equals or capped_band" > $O/run16_parity.txt 2>&1; echo "parity rc $?"; tail -3 $O/run16_parity.txt
s=$(date +%s)
timeout 1200 python -m pytest tests/test_gpu_full_size.py -q -x -k "ecoli or yeast" > $O/run16_full.txt 2>&1; echo "full-size rc $? in $(( $(date +%s) - s )) s"; tail -3 $O/run16_full.txt
run() { # name, env...
  n=$1; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run16_$n.json 2> $O/run16_$n.err; echo "bench $n ($*) rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run16_$n.json') if l.startswith('{"metric"')][-1])
print('   ', d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
}
run ov0 NECAT_EXT_OVERLAP=0
run ov1 NECAT_EXT_OVERLAP=1
run ov1q8 NECAT_EXT_OVERLAP=1 GPU_MAX_HW_QUEUES=8
run ov0q8 NECAT_EXT_OVERLAP=0 GPU_MAX_HW_QUEUES=8
run ov1q8p50 NECAT_EXT_OVERLAP=1 GPU_MAX_HW_QUEUES=8 NECAT_EXT_OVERLAP_PCT=50
run ov1q8p90 NECAT_EXT_OVERLAP=1 GPU_MAX_HW_QUEUES=8 NECAT_EXT_OVERLAP_PCT=90
run ov1q8m100 NECAT_EXT_OVERLAP=1 GPU_MAX_HW_QUEUES=8 NECAT_RC3_MIN=100000
run ov1q8s40 NECAT_EXT_OVERLAP=1 GPU_MAX_HW_QUEUES=8 NECAT_EXT_OVERLAP_SPLIT=40
run ov1q8s65 NECAT_EXT_OVERLAP=1 GPU_MAX_HW_QUEUES=8 NECAT_EXT_OVERLAP_SPLIT=65 NECAT_EXT_OVERLAP_PCT=50
NECAT_TRACE=3 GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-widened --no-pmc > /dev/null 2> $O/run16_trace.err; grep -n "starts on lane\|round" $O/run16_trace.err | tail -90 > $O/run16_trace_tail.txt; tail -75 $O/run16_trace_tail.txt
