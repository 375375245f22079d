# round 5, call 36: which walk kernels the block-level parity cases really launch (rocprofv3 --kernel-trace --stats around single pytest cases)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for k in "edlib_blocks and recompute_16" "edlib_blocks and recompute-" "edlib_blocks and recompute_rows" "edlib_blocks and recompute_quad" "alternative_kernel and RC_WW=2 and not BAND" "alternative_kernel and RC3_MIN=64"; do
  rm -rf $O/p36
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/p36 -o r --output-format csv -- python -m pytest tests/test_gpu_parity.py -q -x -k "$k" > $O/run36.log 2>&1
  echo "== $k: $(tail -1 $O/run36.log)"
  python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r05/p36/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows:
    n = r['Name']
    if 'rcwalk' in n or 'k_walk_wave' in n or 'k_traceback' in n and ', 0,' in n:
        print('   ', r['Calls'], n.split('(')[0][:80])
PY
done
rm -rf $O/p36
