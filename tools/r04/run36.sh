cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
python - > $O/run36_gen.txt 2>&1 <<'PY'
import os, sys, json, time
sys.path.insert(0, os.getcwd())
from necat_amd import synth
g = json.load(open("tests/golden/drosophila_full_reference.json"))["generator"]
rs = synth.simulate_reads(g["genome"], g["coverage"], seed=g["seed"], err=g["err"])
synth.write_volume_dir_cuts("/tmp/dros", rs, g["cuts"])
PY
D=/tmp/dros
OPT="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"
for rep in 1 2 3; do
for mode in "-j 0 -u 1 -i 1" "-j 1 -u 0 -i 0"; do
  rm -f $D/pm*.finished
  s=$(date +%s.%N); NECAT_CLI_TRACE=1 NECAT_GPUS=0 necat_amd/csrc/oc2pm $OPT $mode -t 16 $D /tmp/dros_all > $O/run36_oc2pm.out 2> $O/run36_oc2pm_${rep}_$(echo $mode | cut -c4).err; e=$(date +%s.%N)
  python3 -c "print('rep $rep oc2pm $mode: %.2f s wall' % ($e - $s))"
  grep "\[pm\]" $O/run36_oc2pm_${rep}_$(echo $mode | cut -c4).err | grep "index built\|job done" | tr '\n' ' '; echo
  sleep 2
done; done
md5sum /tmp/dros_all* 2>/dev/null | head -3; ls -la /tmp/dros_all* | head -3
