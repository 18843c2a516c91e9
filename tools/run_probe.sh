cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 300 > gpurun_out/ops.log 2>&1; echo "ops rc=$?"; tail -n 4 gpurun_out/ops.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "golden or batch_invariance" > gpurun_out/parity_sub.log 2>&1; echo "parity rc=$?"; tail -n 6 gpurun_out/parity_sub.log
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-utts 0 > gpurun_out/bench_trim.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/bench_trim.log") if x.startswith("{")]
d=json.loads(l[-1]); print(d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels_ms"].items() if v["ms"]>0.2}); print([(o["precision"], o["value"], o["ms_per_step"], o["stage_ms"]) for o in d["other_precision"]])
PY
