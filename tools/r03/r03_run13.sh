cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k multivolume > $O/run13_mv.txt 2>&1; echo "mv alone rc $?"; tail -3 $O/run13_mv.txt; grep -n "^E  .*hipMalloc" $O/run13_mv.txt | cut -c1-600
rocm-smi --showmeminfo vram 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu > $O/run13_full.txt 2>&1; echo "full rc $?"; tail -3 $O/run13_full.txt; grep -n "^E  .*hipMalloc" $O/run13_full.txt | cut -c1-600
