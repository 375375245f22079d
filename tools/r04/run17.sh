cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
s=$(date +%s)
python - > $O/run17_gen.txt 2>&1 <<'PY'
import os, sys, json, time
sys.path.insert(0, os.getcwd())
from necat_amd import synth
g = json.load(open("tests/golden/drosophila_full_reference.json"))["generator"]
t0 = time.time()
rs = synth.simulate_reads(g["genome"], g["coverage"], seed=g["seed"], err=g["err"])
print("generated in %.0f s" % (time.time() - t0)); t0 = time.time()
synth.write_volume_dir_cuts("/tmp/dros", rs, g["cuts"])
print("written in %.0f s" % (time.time() - t0))
PY
echo "data in $(( $(date +%s) - s )) s"; cat $O/run17_gen.txt
OPT="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"
for mode in "-j 0 -u 1 -i 1" "-j 1 -u 0 -i 0"; do
  rm -f /tmp/dros/pm*.finished
  s=$(date +%s.%N); NECAT_CLI_TRACE=1 NECAT_GPUS=0 necat_amd/csrc/oc2pm $OPT $mode -t 16 /tmp/dros /tmp/dros_all > $O/run17_oc2pm.out 2> $O/run17_oc2pm_$(echo $mode | cut -c4).err; e=$(date +%s.%N)
  python3 -c "print('oc2pm $mode NECAT_GPUS=0: %.2f s wall' % ($e - $s))"
  grep "\[pm\]" $O/run17_oc2pm_$(echo $mode | cut -c4).err | head -60
done
