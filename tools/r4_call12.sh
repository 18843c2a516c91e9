# gpurun --timeout 900 -- 'bash tools/r4_call12.sh'   (round 4: non-temporal stores / slab requests in the MX conv-GEMM: A/B of tuning builds)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for lib in std nt1 nt3 std nt1 nt3; do
  if [ $lib = std ]; then unset EVHIP_LIB; else export EVHIP_LIB=$PWD/emotivoice_amd/csrc/libevhip_$lib.so; fi
  timeout 200 python tools/bench_mxgemm.py --c 128 --reps 8 > gpurun_out/mxgemm_$lib.log 2>&1; echo "mxgemm $lib rc=$?"; grep -E "again" gpurun_out/mxgemm_$lib.log | awk '{print $3,$4,$5,$8}' | tr '\n' ' '; echo
done
for lib in std nt1 nt3 std nt1 nt3; do
  if [ $lib = std ]; then unset EVHIP_LIB; else export EVHIP_LIB=$PWD/emotivoice_amd/csrc/libevhip_$lib.so; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision > gpurun_out/bench_$lib.log 2>&1; echo "bench $lib rc=$?"
  python - $lib <<'PY'
import json, sys
l = [x for x in open("gpurun_out/bench_%s.log" % sys.argv[1]) if x.startswith("{")]
if not l:
    print(open("gpurun_out/bench_%s.log" % sys.argv[1]).read()[-1500:])
else:
    d = json.loads(l[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels_ms"].items() if v["ms"] > 2})
PY
done
