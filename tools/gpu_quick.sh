# scratch runner of the current experiment (gpurun --timeout N -- 'bash tools/gpu_quick.sh'); edit freely, results under gpurun_out/
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "mx" --timeout 600 > gpurun_out/ops_mx.log 2>&1; echo "ops rc=$?"; tail -n 4 gpurun_out/ops_mx.log
( timeout 300 python tools/bench_mxgemm.py --c 128 --valid-shift 6 --dbg 2,0,2 --ks 3,7,11
  timeout 300 python tools/bench_mxgemm.py --c 256 --valid-shift 3 --dbg 2,0,2 --ks 3,11 ) > gpurun_out/mx_latecheck_ab.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/mx_latecheck_ab.txt | cut -c1-160
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision --dump-launches gpurun_out/launches.json > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench.log | cut -c1-300
