# gpurun --timeout 900 -- 'bash tools/r4_call3.sh'   (round 4: persistent tile loop of conv_gemm_mx_kernel)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -k "mx" > gpurun_out/ops_mx.log 2>&1; echo "ops_mx rc=$?"; tail -n 12 gpurun_out/ops_mx.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 500 -k "mx and (golden or batch_invariance or ragged_batch or taps or random_ragged)" > gpurun_out/parity_mx.log 2>&1; echo "parity_mx rc=$?"; tail -n 6 gpurun_out/parity_mx.log
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision --dump-launches gpurun_out/launches.json > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l = [x for x in open("gpurun_out/bench.log") if x.startswith("{")]
if not l:
    print(open("gpurun_out/bench.log").read()[-1500:])
else:
    d = json.loads(l[-1])
    print(d["value"], d["ms_per_step"], d["stage_ms"], d["latency"])
    print({k: v for k, v in d["kernels_ms"].items() if v["ms"] > 0.3})
    L = json.load(open("gpurun_out/launches.json.mx"))
    print(" ".join("%d/%d/%d:%.3f" % (r["N"], r["taps"], r["dil"], r["ms"]) for r in L if r["name"] in ("voc_conv_gemm_mx", "dec_mx_gemm")))
PY
