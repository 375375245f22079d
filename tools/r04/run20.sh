cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O /tmp/keep
s=$(date +%s)
NECAT_TEST_KEEP_VOLS=/tmp/keep timeout 1500 python -m pytest tests/ -q -m gpu -x > $O/run20_suite.txt 2>&1; echo "GPU suite rc $? in $(( $(date +%s) - s )) s"; tail -2 $O/run20_suite.txt
D=/tmp/keep/drosophila
OPT="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"
for rep in 1 2; do
for lend in 1 0; do
for mode in "-j 0 -u 1 -i 1" "-j 1 -u 0 -i 0"; do
  rm -f $D/pm*.finished
  nl=$((1 - lend))
  s=$(date +%s.%N); NECAT_TRACE=2 NECAT_NO_LEND=$nl NECAT_INDEX_OWN_OFFSETS=$nl NECAT_CLI_TRACE=1 NECAT_GPUS=0 necat_amd/csrc/oc2pm $OPT $mode -t 16 $D /tmp/dros_all > $O/run20_oc2pm.out 2> $O/run20_oc2pm_${rep}_${lend}_$(echo $mode | cut -c4).err; e=$(date +%s.%N)
  python3 -c "print('rep $rep oc2pm $mode lend=$lend: %.2f s wall' % ($e - $s))"
  grep "\[pm\]" $O/run20_oc2pm_${rep}_${lend}_$(echo $mode | cut -c4).err | grep "index built\|job done" | tr '\n' ' '; echo
  grep -i "arenas at destroy" $O/run20_oc2pm_${rep}_${lend}_$(echo $mode | cut -c4).err
  sleep 3
done; done; done
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-widened > $O/run20_bench.json 2> $O/run20_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/run20_bench.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['phases_ms_per_step'], d.get('candidates_job0'))
PY
