# round 6, call 11: pair lanes on BASELINE configs[3] at its real size (5.6 Gbp, 3 volumes, 6 pairs) through one oc2pm worker, a pause between the processes (a process that
# starts while the driver still scrubs its predecessor's VRAM waits seconds in its first allocations: run10's walls), stage trace on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_cli_golden.py -q -x -k "lanes or threads" > $O/run11_cli_tests.txt 2>&1; echo "cli tests rc $?"; tail -3 $O/run11_cli_tests.txt
python - > $O/run11_gen.txt 2>&1 <<'PY'
import os, sys, json, time
sys.path.insert(0, os.getcwd())
from necat_amd import synth
g = json.load(open("tests/golden/drosophila_full_reference.json"))["generator"]
rs = synth.simulate_reads(g["genome"], g["coverage"], seed=g["seed"], err=g["err"])
synth.write_volume_dir_cuts("/tmp/dros", rs, g["cuts"])
PY
D=/tmp/dros
OPT="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"
cat $D/vol* > /dev/null
for rep in 1 2 3; do
for ln in 1 2 3; do
  rm -f $D/pm*.finished; sleep 5
  s=$(date +%s.%N); NECAT_CLI_TRACE=1 NECAT_PAIR_LANES=$ln NECAT_GPUS=0 necat_amd/csrc/oc2pm $OPT -j 1 -u 0 -i 0 -t 16 $D /tmp/dros_all > $O/run11_oc2pm_${rep}_$ln.out 2> $O/run11_oc2pm_${rep}_$ln.err; e=$(date +%s.%N)
  python3 -c "print('rep $rep oc2pm -j 1, NECAT_PAIR_LANES=$ln: %.2f s wall' % ($e - $s))"
  sort /tmp/dros_all | md5sum | cut -c1-12
done; done
for rep in 1 2; do
for ln in 1 2 3; do
  rm -f $D/pm*.finished; sleep 5
  s=$(date +%s.%N); NECAT_CLI_TRACE=1 NECAT_PAIR_LANES=$ln NECAT_GPUS=0 necat_amd/csrc/oc2pm $OPT -j 0 -u 1 -i 1 -t 16 $D /tmp/dros_can > $O/run11_oc2pm0_${rep}_$ln.out 2> $O/run11_oc2pm0_${rep}_$ln.err; e=$(date +%s.%N)
  python3 -c "print('rep $rep oc2pm -j 0, NECAT_PAIR_LANES=$ln: %.2f s wall' % ($e - $s))"
  md5sum /tmp/dros_can | cut -c1-12
done; done
