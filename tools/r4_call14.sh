# gpurun --timeout 1200 -- 'bash tools/r4_call14.sh'   (round 4: split-K token-rate GEMMs: op test, parity in every mode, B = 1 latency and forward A/B against the previous library)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for lib in off ffn on off ffn on; do
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision --token-splitk $lib > gpurun_out/bench_$lib.log 2>&1; echo "bench splitk $lib rc=$?"
  python - $lib <<'PY'
import json, sys
l = [x for x in open("gpurun_out/bench_%s.log" % sys.argv[1]) if x.startswith("{")]
if not l:
    print(open("gpurun_out/bench_%s.log" % sys.argv[1]).read()[-1500:])
else:
    d = json.loads(l[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["stage_ms"], d["latency"])
    print({k: v["ms"] for k, v in d["kernels_ms"].items() if k.startswith(("enc", "var", "layer"))})
PY
done
