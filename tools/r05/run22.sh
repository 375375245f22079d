# round 5, call 22: the cut-in-two step as the first / the second context of a process (run21's A/B inside bench.py was 45.8 ms, run19's 37.1), with the kernels' queues
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for o in first second second_closed; do timeout 300 python tools/r05/ab_cut.py $o 2>&1 | grep -v "^$" | tail -9; done > $O/run22_ab.txt; cat $O/run22_ab.txt
rm -rf $O/prof22; timeout 600 rocprofv3 --kernel-trace -d $O/prof22 -o r --output-format csv -- python tools/r05/ab_cut.py second > $O/run22_prof.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r05/prof22/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
print(len(rows), 'kernels; columns', list(rows[0].keys())[:14])
rows.sort(key=lambda r: int(r['Start_Timestamp']))
q = collections.Counter((r['Queue_Id'], r['Kernel_Name'].split('(')[0].replace('void necat::','').replace('necat::','')[:40]) for r in rows[-3000:])
for (qq, k), n in sorted(q.items()): print(qq, k, n)
PY
rm -rf $O/prof22
