# scratch runner of the current experiment (gpurun --timeout N -- 'bash tools/gpu_quick.sh'); edit freely, results under gpurun_out/
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
( echo "# cache-resident problem (270 k rows: 35 MB in + 35 MB out, inside the 256 MB Infinity Cache): what a pair costs when HBM is not the limit"
  timeout 300 python tools/bench_pair_mx.py --ks 3,7,11 --dils 3 --rows 270336 --iters 20 --dbg 0,8,4,0,8,4
  echo "# twice that"
  timeout 300 python tools/bench_pair_mx.py --ks 3 --dils 3 --rows 540672 --iters 20 --dbg 0,8,4 ) > gpurun_out/pair_mx_cache_resident.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/pair_mx_cache_resident.txt | cut -c1-160
