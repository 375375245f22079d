# round 5, call 31: two builds (product without the retired kernel families; libnecat_hip_xcheck.so with them for the alternative-path cases): the whole GPU suite, smoke, short bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
s=$(date +%s)
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/run31_gpu_tests.txt 2>&1; echo "GPU suite rc $? in $(( $(date +%s) - s )) s"; tail -12 $O/run31_gpu_tests.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $O/run31_smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/run31_smoke.txt
timeout 600 python bench.py --no-cpu-baseline --no-widened --no-pmc > $O/run31_bench.json 2> $O/run31_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05/run31_bench.json') if l.startswith('{"metric"')][-1])
print(d['ms_per_step'], d['value'], d['phases_ms_per_step']['extend'], d['roofline']['frac'])
PY
