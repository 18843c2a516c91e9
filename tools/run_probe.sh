cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 300 -k "c64 or mx" > gpurun_out/ops.log 2>&1; echo "ops rc=$?"; tail -n 4 gpurun_out/ops.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "golden and mx" > gpurun_out/parity_sub.log 2>&1; echo "parity rc=$?"; tail -n 4 gpurun_out/parity_sub.log
timeout 300 python tools/bench_c64.py 2>&1 | grep -E "full|no stores at all" 
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision > gpurun_out/bench_trim.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/bench_trim.log") if x.startswith("{")]
d=json.loads(l[-1]); print(d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels_ms"].items() if v["ms"]>1})
PY
