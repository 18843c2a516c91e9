cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 300 -k "mx or attention" > gpurun_out/ops_mx.log 2>&1; echo "ops_mx rc=$?"
tail -n 5 gpurun_out/ops_mx.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "(mx or strict) and (golden or taps or batch_invariance or shortest)" > gpurun_out/parity_mx.log 2>&1; echo "parity_mx rc=$?"
tail -n 30 gpurun_out/parity_mx.log
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-utts 0 --precision mx --no-other-precision --dump-launches gpurun_out/launches_mx.json > gpurun_out/bench_mx.log 2>&1; echo "bench_mx rc=$?"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/bench_mx.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print(d["value"], d["ms_per_step"], d["stage_ms"]); print({k:v for k,v in d["kernels_ms"].items() if v["ms"]>0.3})
else: print(open("gpurun_out/bench_mx.log").read()[-2000:])
PY
