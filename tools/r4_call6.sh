# gpurun --timeout 900 -- 'bash tools/r4_call6.sh'   (round 4: one-tap MX GEMM for the decoder's projections)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -k "mx_one_tap or mx_conv_gemm or mx_plane_set_chain" > gpurun_out/ops_mx.log 2>&1; echo "ops rc=$?"; tail -n 25 gpurun_out/ops_mx.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 500 -k "mx and (golden or batch_invariance or ragged_batch or taps or random_ragged or shortest or long_utterance)" > gpurun_out/parity_mx.log 2>&1; echo "parity rc=$?"; tail -n 6 gpurun_out/parity_mx.log
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision --dump-launches gpurun_out/launches.json > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
l = [x for x in open("gpurun_out/bench.log") if x.startswith("{")]
if not l:
    print(open("gpurun_out/bench.log").read()[-1500:])
else:
    d = json.loads(l[-1])
    print(d["value"], d["ms_per_step"], d["stage_ms"], d["latency"])
    print({k: v for k, v in d["kernels_ms"].items() if v["ms"] > 0.2})
    L = json.load(open("gpurun_out/launches.json.mx"))
    print(" ".join("%s:%.3f" % (r["name"][4:], r["ms"]) for r in L[48:79]))
PY
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity_report.json"))
print({k: (round(v["mel"], 7), round(v["wav_ac"], 7)) for k, v in d.items() if k.startswith("golden") and k.endswith("/mx")})
PY
