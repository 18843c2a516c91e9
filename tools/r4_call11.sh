# gpurun --timeout 1200 -- 'bash tools/r4_call11.sh'   (round 4: 16-row swizzled epilogue scratch in conv_gemm_mx_kernel: all op tests, mx parity, forward A/B against the previous library)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 700 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -x > gpurun_out/ops.log 2>&1; echo "ops rc=$?"; tail -n 6 gpurun_out/ops.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 500 -k "mx and (golden or batch_invariance or ragged_batch or taps or random_ragged or shortest or long_utterance or chunked)" > gpurun_out/parity_mx.log 2>&1; echo "parity rc=$?"; tail -n 4 gpurun_out/parity_mx.log
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/parity_report.json"))
    print({k: (round(v["mel"], 7), round(v["wav_ac"], 7)) for k, v in d.items() if k.startswith("golden") and k.endswith("/mx")})
except Exception as e:
    print("no parity report", e)
PY
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
