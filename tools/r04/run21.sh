cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for cfg in "0 0" "1 0" "0 1"; do
  set -- $cfg
  s=$(date +%s)
  NECAT_INDEX_OWN_OFFSETS=$1 NECAT_NO_LEND=$2 timeout 600 python -m pytest tests/test_gpu_pairs.py tests/test_gpu_shard.py -q -m gpu > $O/run21_own$1_nolend$2.txt 2>&1; echo "own=$1 nolend=$2 rc $? in $(( $(date +%s) - s )) s"; tail -4 $O/run21_own$1_nolend$2.txt; grep -c "hipIpcGetMemHandle failed" $O/run21_own$1_nolend$2.txt
done
