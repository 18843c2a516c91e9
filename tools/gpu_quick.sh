#!/bin/bash
# Scratch runner for one-off GPU experiments:  gpurun --timeout N -- 'bash tools/gpu_quick.sh'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -s -k "fused_mx_resblock_pair and not c64" --timeout 600 > gpurun_out/ops_pair.log 2>&1; echo "ops rc=$?"; grep -E "passed|failed|Error|assert" gpurun_out/ops_pair.log | head -40
timeout 600 python tools/bench_pair_mx.py --ks 3,7,11 --dils 1,5 --dbg 0,16,0,16 2>&1 | grep -v amdgpu.ids > gpurun_out/pair_e5_ab.txt; echo "ab rc=$?"; cut -c1-160 gpurun_out/pair_e5_ab.txt
