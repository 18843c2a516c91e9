# gpurun --timeout 1500 -- 'bash tools/r4_call1.sh'   (round 4, first call: the residual-from-planes default, mx in every parity test, bench A/B)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python -c "import bench; print(bench.csrc_hash())" > gpurun_out/csrc_hash.txt
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 300 -k "mx_residual_from_planes or conv_c64_mx or mx_plane_set_chain or fused_mx" > gpurun_out/ops_mx.log 2>&1; echo "ops_mx rc=$?"; tail -n 3 gpurun_out/ops_mx.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 800 -k "mx" > gpurun_out/parity_mx.log 2>&1; echo "parity_mx rc=$?"; tail -n 15 gpurun_out/parity_mx.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 4 gpurun_out/smoke.log
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-utts 2 --dump-launches gpurun_out/launches.json > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision --mx-residual fp32 > gpurun_out/bench_res32.log 2>&1; echo "bench_res32 rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/bench.log", "gpurun_out/bench_res32.log"):
    l = [x for x in open(f) if x.startswith("{")]
    if not l:
        print(f, open(f).read()[-1500:]); continue
    d = json.loads(l[-1])
    print(f, d["value"], d["ms_per_step"], d["stage_ms"], d["latency"])
    print({k: v for k, v in d["kernels_ms"].items() if v["ms"] > 0.3})
    r = d["roofline"]; print(r["achieved"], r["frac"], r["all_vocoder_convs"])
    for o in d.get("other_precision") or []:
        print(o["precision"], o["value"], o["ms_per_step"])
PY
