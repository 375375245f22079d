# round 6, call 18: SQ counters of the final kernels (one step at a time, so that a kernel's counters are its own), three rocprofv3 --pmc passes -> profiles/r06_sq_counters.json
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --in-flight 1 --no-cpu-baseline --no-widened --no-pmc"
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" \
            "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH"; do
  i=$((i+1)); rm -rf $O/pmc_$i
  timeout 300 rocprofv3 --pmc $pass -d $O/pmc_$i -o r --output-format csv -- $CMD > $O/pmc_$i.log 2>&1; echo "pmc pass $i rc $?"
done
python tools/make_profiles.py counters $O/pmc_1 $O/pmc_2 $O/pmc_3 $O/r06_sq_counters.json; rm -rf $O/pmc_*/
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06/r06_sq_counters.json'))
rows = [(v.get('SQ_INSTS_VALU', 0), v.get('SQ_INSTS_SALU', 0), v.get('launches', 0), v.get('SQ_WAVES', 0), k) for k, v in d.items() if isinstance(v, dict)]
tot = sum(r[0] for r in rows)
for va, sa, n, w, k in sorted(rows, reverse=True)[:14]:
    print("%6.2f G VALU (%4.1f %%) %6.2f G SALU %5d launches %9d waves  %s" % (va / 1e9, 100 * va / tot, sa / 1e9, n, w, k.replace('void necat::', '').replace('necat::', '')[:48]))
print("total VALU %.2f G" % (tot / 1e9))
PY
