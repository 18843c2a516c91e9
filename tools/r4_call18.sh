# gpurun --timeout 900 -- 'bash tools/r4_call18.sh'   (round 4: streamed C = 64 MX conv-GEMM (conv_gemm_mx64_kernel) -- op tests, A/B against the persistent kernel)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -k "c64 or residual_from_planes" > gpurun_out/ops_c64.log 2>&1; echo "ops rc=$?"; tail -n 30 gpurun_out/ops_c64.log | cut -c1-400
timeout 300 python tools/bench_c64.py --ks 7,11 --ab > gpurun_out/c64_ab.log 2>&1; echo "bench rc=$?"; cat gpurun_out/c64_ab.log | tail -30
