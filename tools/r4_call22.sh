# gpurun --timeout 2400 -- 'bash tools/r4_call22.sh'   (round 4, final: full GPU suite, smoke, profiles of the final sources)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 1400 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/full_gpu.log 2>&1; echo "full suite rc=$?"; tail -n 4 gpurun_out/full_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 4 gpurun_out/smoke.log
bash tools/run_profiles.sh
