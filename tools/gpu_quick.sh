#!/bin/bash
# scratch runner: which phases run power-limited (tools/power_probe.py)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python tools/power_probe.py --seconds 3 2>&1 | grep -v amdgpu.ids > gpurun_out/power_probe.txt; echo "rc=$?"; tail -14 gpurun_out/power_probe.txt
