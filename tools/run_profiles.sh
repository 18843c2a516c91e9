# gpurun --timeout 1800 -- 'bash tools/run_profiles.sh'     (bench lines of every mode, kernel trace, PMC traffic / SQ counters of the mx mode;
# then here: python tools/profile_summary.py --tag r4_x)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
bash tools/gpu_check.sh bench prof pmc pmccal pmcsq 2>&1 | tail -n 8
timeout 600 python bench.py --mode ragged --steps 3 --warmup 1 --cpu-utts 0 --no-other-precision > gpurun_out/bench_ragged.log 2>&1; tail -n 1 gpurun_out/bench_ragged.log | cut -c1-300
timeout 600 python bench.py --mode vocoder --steps 3 --warmup 1 --cpu-utts 0 --no-other-precision > gpurun_out/bench_vocoder.log 2>&1; tail -n 1 gpurun_out/bench_vocoder.log | cut -c1-300
