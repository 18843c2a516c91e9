# gpurun --timeout 1200 -- 'bash tools/r4_call19.sh'   (round 4: streamed C = 64 / k = 11 kernel in the engine: ops, mx parity, forward)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -k "c64 or residual_from_planes or plane_set_chain" > gpurun_out/ops_c64.log 2>&1; echo "ops rc=$?"; tail -n 4 gpurun_out/ops_c64.log | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 500 -k "mx and (golden or batch_invariance or ragged_batch or taps or random_ragged or shortest or long_utterance or chunked)" > gpurun_out/parity_mx.log 2>&1; echo "parity rc=$?"; tail -n 4 gpurun_out/parity_mx.log
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/parity_report.json"))
    print({k: (round(v["mel"], 7), round(v["wav_ac"], 7)) for k, v in d.items() if k.startswith("golden") and k.endswith("/mx")})
except Exception as e:
    print("no parity report", e)
PY
for i in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision > gpurun_out/bench_$i.log 2>&1; echo "bench rc=$?"
  python - $i <<'PY'
import json, sys
l = [x for x in open("gpurun_out/bench_%s.log" % sys.argv[1]) if x.startswith("{")]
if not l:
    print(open("gpurun_out/bench_%s.log" % sys.argv[1]).read()[-1500:])
else:
    d = json.loads(l[-1])
    print(d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels_ms"].items() if v["ms"] > 2})
PY
done
