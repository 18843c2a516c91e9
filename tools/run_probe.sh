cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
EVHIP_LIB=emotivoice_amd/csrc/libevhip_mxabl.so timeout 300 python tools/bench_mxgemm.py --c 128 --ks 3,11 --reps 10 2>&1 | grep -v amdgpu.ids > gpurun_out/mxgemm_abl2_c128.txt
cat gpurun_out/mxgemm_abl2_c128.txt
