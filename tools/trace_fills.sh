cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/prof_kt
rocprofv3 --kernel-trace -d gpurun_out/prof_kt -o r --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-widened > gpurun_out/prof_kt.log 2>&1
python - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/prof_kt/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
fills=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']), i) for i,r in enumerate(rows) if 'fillBuffer' in r['Kernel_Name']]
print('fills', len(fills), 'total ms', sum(d for d,_ in fills)/1e6)
big=sorted(fills, reverse=True)[:24]
for d,i in big:
    prev=rows[i-1]['Kernel_Name'][:40] if i else ''
    nxt=rows[i+1]['Kernel_Name'][:40] if i+1<len(rows) else ''
    r=rows[i]
    print('%8.1f us  grid %s wg %s  queue %s | prev %s | next %s' % (d/1e3, r.get('Grid_Size',r.get('Grid_Size_X')), r.get('Workgroup_Size',r.get('Workgroup_Size_X')), r.get('Queue_Id'), prev, nxt))
hist=collections.Counter(min(9,int(d/1e3//25)) for d,_ in fills)
print(sorted(hist.items()))
PY
rm -rf gpurun_out/prof_kt
