# round 5, call 11: k_rcwalk3 on 16-diagonal records (walker state in LDS, 72 registers, 7 waves per SIMD) against 32-diagonal ones and k_rcwalk2w: equality + time alone, then in the bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 300 tools/rcwalk_microbench > $O/run11_micro.txt 2>&1; echo "microbench rc $?"; grep -v "^---" $O/run11_micro.txt | head -34
timeout 300 tools/rcwalk_microbench 0.22 15 2>&1 | grep -E "blocks:|==|lean  |lean again|ops kept|prio|half|ONE|one workgroup" > $O/run11_micro_22_15.txt; cat $O/run11_micro_22_15.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "edlib_blocks or alternative_kernel" > $O/run11_parity.txt 2>&1; echo "parity rc $?"; tail -2 $O/run11_parity.txt
run() { n=$1; shift
  env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened --no-pmc > $O/run11_$n.json 2> $O/run11_$n.err; echo "bench $n ($*) rc $?"
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r05/run11_$n.json') if l.startswith('{"metric"')][-1])
print('   ', d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['phases_ms_per_step']['rcwalk_kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
}
run b32 NECAT_RC3_BAND=32
run b16 NECAT_RC3_BAND=16
run b16m100 NECAT_RC3_BAND=16 NECAT_RC3_MIN=100000
run b16m60 NECAT_RC3_BAND=16 NECAT_RC3_MIN=60000
run b16all NECAT_RC3_BAND=16 NECAT_RC_WW=2
run b32b NECAT_RC3_BAND=32
run b16b NECAT_RC3_BAND=16
