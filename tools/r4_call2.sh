# gpurun --timeout 1200 -- 'bash tools/r4_call2.sh'   (round 4: the two-group persistent kernels -- op tests incl. bit-identity vs the lock-step kernels, parity, bench)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python -c "import bench; print(bench.csrc_hash())" > gpurun_out/csrc_hash.txt
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 300 -k "mx_residual_from_planes or conv_c64_mx or mx_plane_set_chain or fused_mx" > gpurun_out/ops_mx.log 2>&1; echo "ops_mx rc=$?"; tail -n 12 gpurun_out/ops_mx.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 500 -k "mx and (golden or batch_invariance or ragged_batch or taps or random_ragged or chunked_vocoding)" > gpurun_out/parity_mx.log 2>&1; echo "parity_mx rc=$?"; tail -n 12 gpurun_out/parity_mx.log
timeout 400 python bench.py --steps 10 --warmup 3 --cpu-utts 0 --no-other-precision --dump-launches gpurun_out/launches.json > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/bench.log",):
    l = [x for x in open(f) if x.startswith("{")]
    if not l:
        print(f, open(f).read()[-1500:]); continue
    d = json.loads(l[-1])
    print(f, d["value"], d["ms_per_step"], d["stage_ms"], d["latency"])
    print({k: v for k, v in d["kernels_ms"].items() if v["ms"] > 0.3})
L = json.load(open("gpurun_out/launches.json.mx"))
print(" ".join("%s:%d/%d:%.3f" % (r["name"][4:12], r["taps"], r["dil"], r["ms"]) for r in L if r["name"].startswith("voc") and r["N"] <= 64))
PY
