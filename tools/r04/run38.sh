cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
s=$(date +%s)
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/r04_final_gpu_tests.txt 2>&1; echo "GPU suite rc $? in $(( $(date +%s) - s )) s"; tail -6 $O/r04_final_gpu_tests.txt | head -2
python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke.txt 2>&1; echo "smoke rc $?"; tail -1 $O/r04_smoke.txt
