cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
for n in 200000 2000; do for v in 0 1 2; do echo "== N=$n NECAT_WALK=$v"; NECAT_WALK=$v NECAT_BATCH_CHUNK=200000 timeout 300 python tools/bench_myers.py $n 2>&1 | tail -1; done; done | tee $O/ab_walk.txt
