# gpurun --timeout 900 -- 'bash tools/r4_call4.sh'   (round 4: fused C = 64 / k = 3 MX pair)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -k "fused_mx_resblock_pair_c64" > gpurun_out/ops_mx.log 2>&1; echo "ops rc=$?"; tail -n 25 gpurun_out/ops_mx.log
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
    print(" ".join("%s:%d/%d:%.3f" % (r["name"][4:16], r["taps"], r["dil"], r["ms"]) for r in L if r["name"].startswith("voc") and r["N"] == 64))
PY
