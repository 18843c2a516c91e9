# gpurun --timeout 1800 -- 'bash tools/run_r3_profiles.sh'     (round 3: host pipeline + front end, kernel trace, PMC traffic / SQ counters of the mx mode)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python tools/bench_frontend.py --lines 40000 --workers 1,4,16,64 --json gpurun_out/frontend.json > gpurun_out/frontend.log 2>&1; tail -n 5 gpurun_out/frontend.log | cut -c1-300
timeout 600 python bench.py --mode pipeline --steps 20 --warmup 3 --frontend-workers 1 > gpurun_out/pipeline_w1.log 2>&1; tail -n 1 gpurun_out/pipeline_w1.log | cut -c1-1200
timeout 600 python bench.py --mode pipeline --steps 20 --warmup 3 --frontend-workers 8 > gpurun_out/pipeline_w8.log 2>&1; tail -n 1 gpurun_out/pipeline_w8.log | cut -c1-1200
bash tools/gpu_check.sh prof pmc pmccal pmcsq 2>&1 | tail -n 5
