# gpurun --timeout 900 -- 'bash tools/r4_call13.sh'   (round 4: residual-from-planes epilogues request pass p + 1 while pass p computes)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 500 -x -k "mx" > gpurun_out/ops_mx.log 2>&1; echo "ops rc=$?"; tail -n 4 gpurun_out/ops_mx.log | cut -c1-300
for lib in base new base new; do
  if [ $lib = base ]; then export EVHIP_LIB=$PWD/emotivoice_amd/csrc/libevhip_base.so; else unset EVHIP_LIB; fi
  timeout 200 python tools/bench_mxgemm.py --c 128 --reps 8 > gpurun_out/mxgemm_$lib.log 2>&1; echo "mxgemm $lib rc=$?"; grep -E "again" gpurun_out/mxgemm_$lib.log | awk '{print $3,$4,$5,$8}' | tr '\n' ' '; echo
done
for lib in base new base new; do
  if [ $lib = base ]; then export EVHIP_LIB=$PWD/emotivoice_amd/csrc/libevhip_base.so; else unset EVHIP_LIB; fi
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
