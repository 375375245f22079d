# round 5, call 37: the two-lane scheduler under repetition - the several-batches cases in every lane mode and the full-size md5s (yeast: four batches on two lanes;
# E. coli / yeast with one batch cut in two), eight times over
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
: > $O/run37_stress.txt
for i in 1 2 3 4 5 6 7 8; do
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -q -x -k "several_batches or ecoli or yeast" >> $O/run37_stress.txt 2>&1; echo "iteration $i rc $?"
done
grep -c passed $O/run37_stress.txt; grep "passed\|failed\|error" $O/run37_stress.txt | sort | uniq -c
