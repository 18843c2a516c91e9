cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -k "layernorm or ln_" > gpurun_out/q_ops.log 2>&1; echo "ops rc=$?"; tail -n 2 gpurun_out/q_ops.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "golden and (mx or strict) and (tiny or n33 or hot_zdc or stress)" > gpurun_out/q_parity.log 2>&1; echo "parity rc=$?"; tail -n 2 gpurun_out/q_parity.log
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision > gpurun_out/q_bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/q_bench.log") if x.startswith("{")]
d=json.loads(l[-1]); print(d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels_ms"].items() if k in ("voc_conv_post","gauss_upsample","layernorm")}, d["stage_ms"])
PY
