# scratch runner of the current experiment (gpurun --timeout N -- 'bash tools/gpu_quick.sh'); edit freely, results under gpurun_out/
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "fused_mx_resblock_pair" --timeout 600 > gpurun_out/ops_pair.log 2>&1; echo "ops rc=$?"; tail -n 4 gpurun_out/ops_pair.log
timeout 900 python tools/bench_pair_mx.py --ks 3,7,11 --dils 1,5 --dbg 0,8,0,8 > gpurun_out/pair_mx_trim_ab.txt 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids gpurun_out/pair_mx_trim_ab.txt | cut -c1-160
