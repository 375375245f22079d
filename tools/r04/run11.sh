cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "edlib_blocks or alternative_kernel or onc_align or ultra" > $O/run11_parity.txt 2>&1; echo "parity rc $?"; tail -3 $O/run11_parity.txt
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_asmpm.py tests/test_gpu_zz_asm_align.py tests/test_gpu_cns.py -q -m gpu -x -k "not drosophila and not multivol" > $O/run11_full.txt 2>&1; echo "full-size rc $?"; tail -3 $O/run11_full.txt
for m in 1 0 1 0; do
  NECAT_RC_PREFETCH=$m timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-widened > $O/run11_bench_pf$m.json 2> $O/run11_bench_pf$m.err
  python3 -c "
import json
d=json.loads(open('$O/run11_bench_pf$m.json').read().strip().splitlines()[-1]); print('prefetch', $m, d['ms_per_step'], d['phases_ms_per_step'])"
done
