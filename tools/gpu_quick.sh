#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
( EVHIP_LIB=emotivoice_amd/csrc/libevhip_mxabl.so timeout 600 python tools/bench_mxgemm.py --c 128 --valid-shift 6 --ks 3,7,11 --reps 10
  EVHIP_LIB=emotivoice_amd/csrc/libevhip_mxabl.so timeout 600 python tools/bench_mxgemm.py --c 256 --valid-shift 3 --ks 3,11 --reps 10 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/mx_abl_l2.txt; echo "rc=$?"; grep -E "full|L2|one-chunk|no epilogue" gpurun_out/mx_abl_l2.txt | cut -c1-200
