#!/bin/bash
# Scratch runner for one-off GPU experiments:  gpurun --timeout N -- 'bash tools/gpu_quick.sh'
# Edit the body for the experiment at hand (results under gpurun_out/, copy what should be judged into profiles/).  The stable entry points are
# tools/gpu_check.sh (tests / smoke / bench / profiles by stage name) and tools/run_profiles.sh (the full profile set).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench_default.log | cut -c1-200
