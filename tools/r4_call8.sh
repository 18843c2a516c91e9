# gpurun --timeout 900 -- 'bash tools/r4_call8.sh'   (round 4: MRF partial plane sets + LayerNorm -> plane sets: first device run, A/B against the fp32 flows)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -k "mrf_partial or layernorm_writes or residual_from_planes or plane_set_chain or one_tap or layernorm_and_head" > gpurun_out/ops_new.log 2>&1; echo "ops rc=$?"; tail -n 25 gpurun_out/ops_new.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 500 -k "mx and (golden or batch_invariance or ragged_batch or taps or random_ragged or shortest or long_utterance or chunked)" > gpurun_out/parity_mx.log 2>&1; echo "parity rc=$?"; tail -n 8 gpurun_out/parity_mx.log
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/parity_report.json"))
    print({k: (round(v["mel"], 7), round(v["wav_ac"], 7)) for k, v in d.items() if k.startswith("golden") and k.endswith("/mx")})
except Exception as e:
    print("no parity report", e)
PY
for cfg in new old new old; do
  if [ $cfg = old ]; then X="--mx-mrf fp32 --decoder-ln fp32"; else X=""; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision $X --dump-launches gpurun_out/launches_$cfg.json > gpurun_out/bench_$cfg.log 2>&1; echo "bench $cfg rc=$?"
  python - $cfg <<'PY'
import json, sys
l = [x for x in open("gpurun_out/bench_%s.log" % sys.argv[1]) if x.startswith("{")]
if not l:
    print(open("gpurun_out/bench_%s.log" % sys.argv[1]).read()[-1500:])
else:
    d = json.loads(l[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["stage_ms"], d.get("latency"))
    print({k: v["ms"] for k, v in d["kernels_ms"].items() if v["ms"] > 0.2})
PY
done
