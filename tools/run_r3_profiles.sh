# gpurun --timeout 1800 -- 'bash tools/run_r3_profiles.sh'     (round 3: bench lines of every mode, host pipeline + front end, kernel trace, PMC traffic / SQ counters of the mx mode)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
EV_DUMP_LAUNCHES=gpurun_out/launches.json timeout 900 python bench.py --steps 10 --warmup 3 --cpu-utts 4 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench.log | cut -c1-400
timeout 600 python bench.py --mode ragged --steps 3 --warmup 1 --cpu-utts 0 --no-other-precision > gpurun_out/bench_ragged.log 2>&1; tail -n 1 gpurun_out/bench_ragged.log | cut -c1-300
timeout 600 python bench.py --mode vocoder --steps 3 --warmup 1 --cpu-utts 0 --no-other-precision > gpurun_out/bench_vocoder.log 2>&1; tail -n 1 gpurun_out/bench_vocoder.log | cut -c1-300
timeout 600 python bench.py --mode pipeline --steps 20 --warmup 3 --frontend-workers 1 > gpurun_out/pipeline_w1.log 2>&1; tail -n 1 gpurun_out/pipeline_w1.log | cut -c1-700
bash tools/gpu_check.sh prof pmc pmccal pmcsq 2>&1 | tail -n 5
