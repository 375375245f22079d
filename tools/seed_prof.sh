# phase breakdown of k_seed_eval (cycles of lane 0, NECAT_SEED_PROF build of the same sources); run on the GPU box
cd ${GRAFT_REPO_ROOT:-.}/necat_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -disable-promote-alloca-to-lds -DNECAT_SEED_PROF -shared -o libnecat_hip_prof.so necat_hip.hip && cd ../.. && \
NECAT_HIP_LIB=$PWD/necat_amd/csrc/libnecat_hip_prof.so python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-widened 2>&1 | grep "seed prof" | tail -12
