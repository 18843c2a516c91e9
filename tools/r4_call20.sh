cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for i in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision --dump-launches gpurun_out/launches_$i.json > gpurun_out/bench_$i.log 2>&1; echo "bench rc=$?"
  python - $i <<'PY'
import json, sys
l = [x for x in open("gpurun_out/bench_%s.log" % sys.argv[1]) if x.startswith("{")]
d = json.loads(l[-1])
print(d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels_ms"].items() if v["ms"] > 2})
L = json.load(open("gpurun_out/launches_%s.json.mx" % sys.argv[1]))
print(" ".join("%d/%d:%.3f" % (r["taps"], r["dil"], r["ms"]) for r in L if r["name"] == "voc_conv_c64_mx"))
PY
done
