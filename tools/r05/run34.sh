# round 5, call 34: BASELINE configs[3] at its real size (140 Mb x 40 = 5.6 Gbp, volumes 2.0 / 2.0 / 1.6 Gbp) through ONE oc2pm worker: two extension lanes against one
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
python - > $O/run34_gen.txt 2>&1 <<'PY'
import os, sys, json, time
sys.path.insert(0, os.getcwd())
from necat_amd import synth
g = json.load(open("tests/golden/drosophila_full_reference.json"))["generator"]
rs = synth.simulate_reads(g["genome"], g["coverage"], seed=g["seed"], err=g["err"])
synth.write_volume_dir_cuts("/tmp/dros", rs, g["cuts"])
PY
D=/tmp/dros
OPT="-k 15 -z 20 -q 500 -b 2000 -s 3 -n 500 -a 1000 -d 0.25 -e 0.5 -m 500"
for rep in 1 2; do
for ov in 1 0; do
  rm -f $D/pm*.finished
  s=$(date +%s.%N); NECAT_EXT_OVERLAP=$ov NECAT_GPUS=0 necat_amd/csrc/oc2pm $OPT -j 1 -u 0 -i 0 -t 16 $D /tmp/dros_all > $O/run34_oc2pm.out 2> $O/run34_oc2pm_${rep}_$ov.err; e=$(date +%s.%N)
  python3 -c "print('rep $rep oc2pm -j 1, NECAT_EXT_OVERLAP=$ov: %.2f s wall' % ($e - $s))"
  md5sum /tmp/dros_all | cut -c1-12; sort /tmp/dros_all | md5sum | cut -c1-12
done; done
python3 -c "
import json; g=json.load(open('tests/golden/drosophila_full_reference.json')); print('golden sorted md5', g['m4_text_sorted_md5'][:12], g['m4_records'])"
wc -l /tmp/dros_all
