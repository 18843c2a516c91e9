cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
EV_DUMP_LAUNCHES=gpurun_out/launches.json timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench.log | cut -c1-300
timeout 600 python bench.py --mode ragged --steps 3 --warmup 1 --cpu-utts 0 --no-other-precision > gpurun_out/bench_ragged.log 2>&1; tail -n 1 gpurun_out/bench_ragged.log | cut -c1-200
timeout 600 python bench.py --mode vocoder --steps 3 --warmup 1 --cpu-utts 0 --no-other-precision > gpurun_out/bench_vocoder.log 2>&1; tail -n 1 gpurun_out/bench_vocoder.log | cut -c1-200
bash tools/gpu_check.sh prof 2>&1 | tail -n 3
