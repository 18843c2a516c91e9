#!/bin/bash
# Scratch runner for one-off GPU experiments:  gpurun --timeout N -- 'bash tools/gpu_quick.sh'
# Edit the body for the experiment at hand (results under gpurun_out/, copy what should be judged into profiles/).  The stable entry points are
# tools/gpu_check.sh (tests / smoke / bench / profiles by stage name) and tools/run_profiles.sh (the full profile set).
# Round 6's experiments, as they were run (tuning builds travel with the snapshot: build them here first):
#   python emotivoice_amd/csrc/build.py --variant mxt EV_MXT          # chip-wide timeline of conv_gemm_mx_kernel (profiles/r6_e_mx_timeline.txt)
#   python emotivoice_amd/csrc/build.py --variant ptime EV_PAIR_TIMING  # per-phase ticks of resblock_pair_c32_e5_kernel (profiles/r6_b_pair_e5_ab.txt)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python tools/bench_pair_mx.py --ks 3,7,11 --dils 1,5 --dbg 0,16,32,0,16,32 2>&1 | grep -v amdgpu.ids > gpurun_out/pair_e5_ab.txt; echo "ab rc=$?"; cut -c1-160 gpurun_out/pair_e5_ab.txt
if [ -f emotivoice_amd/csrc/libevhip_mxt.so ]; then
  EVHIP_LIB=emotivoice_amd/csrc/libevhip_mxt.so timeout 600 python tools/bench_mxgemm.py --c 128 --valid-shift 6 --ks 3,11 --reps 5 --timeline 2>&1 | grep -v amdgpu.ids > gpurun_out/mx_timeline.txt; echo "timeline rc=$?"
  grep -E "timeline:|store drain|wave skew|per wave|epilogue per wave" gpurun_out/mx_timeline.txt | cut -c1-330
fi
