cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
NECAT_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "rc $?"; tail -3 $O/bench_2rank.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02/bench_2rank.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['n_gpus'], d['scaling'], d['config'].get('parallelism'))
print(d.get('multi_gpu'))
PY
