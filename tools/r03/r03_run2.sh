# round 3, second GPU pass: non-temporal loads in the walk (A/B), SQ counter breakdown of the list-A DP kernel on 200 k synthetic full blocks
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
python -c "from necat_amd import build; build.build_hip()"
timeout 300 python -m pytest tests/test_gpu_pairs.py -m gpu -q --timeout 600 -k shares 2>&1 | tail -3
for W in 0 3 4 1; do echo "== NECAT_WALK=$W"; NECAT_WALK=$W timeout 120 python tools/bench_myers.py 200000 2>&1 | tail -2; done > $O/ab_walk_nt.txt 2>&1
cat $O/ab_walk_nt.txt
for W in 0 3; do
  NECAT_WALK=$W timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-widened > $O/ab_walkbench_$W.json 2> $O/ab_walkbench_$W.err
  python - <<PY
import json
d=json.loads(open('$O/ab_walkbench_$W.json').read().strip().splitlines()[-1])
print('WALK=$W', d['ms_per_step'], d['config']['overlaps_per_step'], d['phases_ms_per_step'])
PY
done
CMD="python tools/bench_myers.py 200000"
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" \
            "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_IFETCH SQ_ACTIVE_INST_MISC" \
            "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LEVEL_WAVES SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
            "GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1)); rm -rf $O/pmc_$i
  timeout 200 rocprofv3 --pmc $pass -d $O/pmc_$i -o r --output-format csv -- $CMD > $O/pmc_$i.log 2>&1; echo "pmc pass $i rc $?"
done
python tools/make_profiles.py counters $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/pmc_4 $O/pmc_5 $O/r03_sq_counters_myers_micro.json; rm -rf $O/pmc_*/
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/r03_sq_counters_myers_micro.json'))
for k,v in d.items():
    if 'k_myers_coop' in k or 'k_traceback' in k:
        print(k[:70]); print({a:b for a,b in v.items()})
PY
