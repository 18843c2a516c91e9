cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for lib in libevhip.so libevhip_mxnoprio.so; do
  echo "== $lib"
  EVHIP_LIB=emotivoice_amd/csrc/$lib timeout 200 python tools/bench_mxgemm.py --c 128 --reps 10 2>&1 | grep again
  EVHIP_LIB=emotivoice_amd/csrc/$lib timeout 200 python tools/bench_mxgemm.py --c 256 --rows 264192 --reps 10 2>&1 | grep again
done > gpurun_out/mxgemm_noprio.txt
cat gpurun_out/mxgemm_noprio.txt
for lib in libevhip.so libevhip_mxnoprio.so; do
  EVHIP_LIB=emotivoice_amd/csrc/$lib timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --precision mx --no-other-precision > gpurun_out/bench_$lib.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/bench_$lib.log") if x.startswith("{")]
d=json.loads(l[-1]); print("$lib", d["value"], d["ms_per_step"], {k:v["ms"] for k,v in d["kernels_ms"].items() if v["ms"]>1})
PY
done
