cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O /tmp/keep
s=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py -q -m gpu -x -k "index or shard" > $O/run18_index.txt 2>&1; echo "index/shard rc $? in $(( $(date +%s) - s )) s"; tail -2 $O/run18_index.txt
s=$(date +%s)
NECAT_TEST_KEEP_VOLS=/tmp/keep timeout 900 python -m pytest tests/test_gpu_full_size.py -q -m gpu -x -k "drosophila" > $O/run18_dros.txt 2>&1; echo "dros rc $? in $(( $(date +%s) - s )) s"; tail -2 $O/run18_dros.txt
D=$(ls -d /tmp/keep/*dros* | head -1); echo "volumes in $D"
OPT="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"
for own in 0 1; do
for mode in "-j 0 -u 1 -i 1" "-j 1 -u 0 -i 0"; do
  rm -f $D/pm*.finished
  s=$(date +%s.%N); NECAT_INDEX_OWN_OFFSETS=$own NECAT_CLI_TRACE=1 NECAT_GPUS=0 necat_amd/csrc/oc2pm $OPT $mode -t 16 $D /tmp/dros_all > $O/run18_oc2pm.out 2> $O/run18_oc2pm_${own}_$(echo $mode | cut -c4).err; e=$(date +%s.%N)
  python3 -c "print('oc2pm $mode NECAT_GPUS=0 own_offsets=$own: %.2f s wall' % ($e - $s))"
  grep "\[pm\]" $O/run18_oc2pm_${own}_$(echo $mode | cut -c4).err | grep -v "records\|candidates\|mapped" | head -20
done; done
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-widened > $O/run18_bench.json 2> $O/run18_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/run18_bench.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'], d.get('candidates_job0'))
PY
