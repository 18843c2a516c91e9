#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
( EVHIP_LIB=emotivoice_amd/csrc/libevhip_mxt.so timeout 600 python tools/bench_mxgemm.py --c 128 --valid-shift 6 --ks 3,7,11 --reps 10 --dbg 4,0,4,0 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/mx_prio2.txt; echo "rc=$?"; grep -v "conv2 " gpurun_out/mx_prio2.txt | cut -c1-200
