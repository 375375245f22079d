cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
python -c "from necat_amd import build; build.build_hip()"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "edlib or coop_equals" 2>&1 | tail -3
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-widened"
rm -rf $O/prof_kt; rocprofv3 --kernel-trace -d $O/prof_kt -o r --output-format csv -- $CMD > $O/prof_kt.log 2>&1
python tools/make_profiles.py timeline $O/prof_kt $O/r03_round_timeline_wavewalk.txt "one bench step kernel by kernel (rocprofv3 --kernel-trace, $CMD; last step), NECAT_WALK_WAVE=12288 NECAT_TAIL_FUSED=512"
rm -rf $O/prof_kt
wc -l $O/r03_round_timeline_wavewalk.txt
