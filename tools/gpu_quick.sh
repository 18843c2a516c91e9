#!/bin/bash
# Scratch runner for one-off GPU experiments:  gpurun --timeout N -- 'bash tools/gpu_quick.sh'
# Edit the body for the experiment at hand (results under gpurun_out/, copy what should be judged into profiles/).  The stable entry points are
# tools/gpu_check.sh (tests / smoke / bench / profiles by stage name) and tools/run_profiles.sh (the full profile set).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -q --timeout 600 > gpurun_out/api.log 2>&1; echo "api rc=$?"; tail -n 4 gpurun_out/api.log
