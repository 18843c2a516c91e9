# gpurun --timeout 900 -- 'bash tools/r4_call9.sh'   (round 4: persistent conv_gemm_mx_kernel with the next tile's first requests under the epilogue)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 500 -x -k "persistent or long_tile_list" > gpurun_out/ops_persist.log 2>&1; echo "ops rc=$?"; tail -n 30 gpurun_out/ops_persist.log | cut -c1-300
timeout 300 python tools/bench_mxgemm.py --c 128 --dbg 0,8,0,8 > gpurun_out/mxgemm_c128.log 2>&1; echo "mxgemm rc=$?"; grep -E "dbg|full" gpurun_out/mxgemm_c128.log
timeout 300 python tools/bench_mxgemm.py --c 256 --dbg 0,8,0,8 > gpurun_out/mxgemm_c256.log 2>&1; echo "mxgemm rc=$?"; grep -E "dbg|full" gpurun_out/mxgemm_c256.log
for lib in base new base new; do
  if [ $lib = base ]; then export EVHIP_LIB=$PWD/emotivoice_amd/csrc/libevhip_base.so; else unset EVHIP_LIB; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision --dump-launches gpurun_out/launches_$lib.json > gpurun_out/bench_$lib.log 2>&1; echo "bench $lib rc=$?"
  python - $lib <<'PY'
import json, sys
l = [x for x in open("gpurun_out/bench_%s.log" % sys.argv[1]) if x.startswith("{")]
if not l:
    print(open("gpurun_out/bench_%s.log" % sys.argv[1]).read()[-1500:])
else:
    d = json.loads(l[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["stage_ms"])
    print({k: v["ms"] for k, v in d["kernels_ms"].items() if v["ms"] > 0.2})
PY
done
