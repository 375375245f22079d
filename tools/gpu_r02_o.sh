cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "index or full or cand" 2>&1 | tail -3
python tools/bench_index.py
cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/px -o r --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_index.py > /dev/null 2>&1; python - <<'PY'
import csv,glob
f=glob.glob('/tmp/px/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:50], r['Calls'], r['AverageNs'])
PY
