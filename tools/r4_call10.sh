# gpurun --timeout 600 -- 'bash tools/r4_call10.sh'   (round 4: persistent conv_gemm_mx_kernel -- what costs it: the 16-row epilogue? dbg 72 = one block per tile with that epilogue)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 300 python tools/bench_mxgemm.py --c 128 --reps 8 --dbg 8,72,0,8,72,0 > gpurun_out/mxgemm_c128.log 2>&1; echo "mxgemm rc=$?"; grep -E "dbg" gpurun_out/mxgemm_c128.log
