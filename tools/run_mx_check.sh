cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 300 -k "mx" > gpurun_out/ops_mx.log 2>&1; echo "ops_mx rc=$?"
tail -n 25 gpurun_out/ops_mx.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "mx and (golden or taps or batch_invariance or shortest)" > gpurun_out/parity_mx.log 2>&1; echo "parity_mx rc=$?"
tail -n 30 gpurun_out/parity_mx.log
EV_DUMP_LAUNCHES=gpurun_out/launches_mx.json timeout 600 python bench.py --steps 5 --warmup 2 --cpu-utts 0 --precision mx --no-other-precision > gpurun_out/bench_mx.log 2>&1; echo "bench_mx rc=$?"
tail -c 1500 gpurun_out/bench_mx.log
