# round 2, GPU call B: multi-rank path (2-3 ranks on device 0, IPC transport), full-size configs[2] parity, microbench 2, bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_shard.py -x -q > $O/pytest_shard.log 2>&1; echo "pytest shard rc $?"; tail -25 $O/pytest_shard.log | cut -c1-300
./tools/valu_microbench2 > $O/valu_microbench2.txt 2>&1; cat $O/valu_microbench2.txt
NECAT_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-widened > $O/bench_2rank_onedev.json 2> $O/bench_2rank_onedev.err; echo "bench 2-rank rc $?"; cut -c1-1500 $O/bench_2rank_onedev.json; tail -5 $O/bench_2rank_onedev.err | cut -c1-300
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_shard.py > $O/pytest_b.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_b.log | cut -c1-300
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_b.json 2> $O/bench_b.err; echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02/bench_b.json'))
for k in ('value','ms_per_step','phases_ms_per_step','candidates_job0','oc2pmov_cold_start','cpu_baseline'): print(k, d.get(k))
PY
