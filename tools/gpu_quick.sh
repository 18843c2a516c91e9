#!/bin/bash
# Scratch runner for one-off GPU experiments:  gpurun --timeout N -- 'bash tools/gpu_quick.sh'
# Edit the body for the experiment at hand (results under gpurun_out/, copy what should be judged into profiles/).  The stable entry points are
# tools/gpu_check.sh (tests / smoke / bench / profiles by stage name) and tools/run_profiles.sh (the full profile set).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
( for rows in 262144 264192 266240 393216; do echo "## rows $rows = $((rows/256*2)) blocks of 256 x 128"; timeout 300 python tools/bench_mxgemm.py --c 256 --rows $rows --ks 3,11 --reps 20 2>&1 | grep -v amdgpu.ids | grep -v "full  "; done ) > gpurun_out/mx_tail_round.txt 2>&1; echo "rc=$?"; cut -c1-150 gpurun_out/mx_tail_round.txt
