#!/bin/bash
# Scratch runner for one-off GPU experiments:  gpurun --timeout N -- 'bash tools/gpu_quick.sh'
# Edit the body for the experiment at hand (results under gpurun_out/, copy what should be judged into profiles/).  The stable entry points are
# tools/gpu_check.sh (tests / smoke / bench / profiles by stage name) and tools/run_profiles.sh (the full profile set).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "mx" --timeout 600 > gpurun_out/ops_mx.log 2>&1; echo "ops rc=$?"; tail -n 3 gpurun_out/ops_mx.log
( timeout 400 python tools/bench_mxgemm.py --c 128 --valid-shift 6 --dbg 2,0,2,0 --ks 3,7,11 --reps 10
  timeout 400 python tools/bench_mxgemm.py --c 256 --valid-shift 3 --dbg 2,0,2,0 --ks 3,11 --reps 10 ) 2>&1 | grep -v amdgpu.ids | grep -E "conv1|conv2pl" | grep -v "full " > gpurun_out/mx_epi_static_ab.txt; echo "ab rc=$?"; cut -c1-150 gpurun_out/mx_epi_static_ab.txt
